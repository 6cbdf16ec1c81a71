/*
 * hsrle_oracle.c -- CPU restatement of the rleX_extreme codecs.  TEST INFRASTRUCTURE ONLY.
 * See hsrle_oracle.h for the parity status (PINNED against oracle/_ref and tests/golden/).
 *
 * Written from the format specification in SURVEY.md Appendix A; every function cites the
 * reference location it follows (paths relative to the reference's src/).  The structure is
 * deliberately different from the reference: one scalar "enumerate runs, decide, emit"
 * encoder per family instead of SIMD bodies instantiated by #include.
 */
#include "hsrle_oracle.h"

#include <stddef.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* byte sink                                                                                  */

typedef struct {
  uint8_t *o;
  size_t at;
} sink_t;

static void put8(sink_t *s, uint32_t v) { s->o[s->at++] = (uint8_t)v; }
static void put16(sink_t *s, uint32_t v) { put8(s, v); put8(s, v >> 8); }
static void put32(sink_t *s, uint32_t v) { put16(s, v); put16(s, v >> 16); }
static void putn(sink_t *s, const uint8_t *p, size_t n) { if (n) memcpy(s->o + s->at, p, n); s->at += n; }
static void putzeros(sink_t *s, size_t n) { memset(s->o + s->at, 0, n); s->at += n; }
static void patch32(uint8_t *o, size_t at, uint32_t v) { o[at] = (uint8_t)v; o[at + 1] = (uint8_t)(v >> 8); o[at + 2] = (uint8_t)(v >> 16); o[at + 3] = (uint8_t)(v >> 24); }
static uint32_t get16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
static uint32_t get32(const uint8_t *p) { return get16(p) | (get16(p + 2) << 16); }

/* rle_compress_bounds: rle8_extreme_cpu.c:22-28 */
uint32_t hso_compress_bounds(uint32_t inSize)
{
  if (inSize > (1u << 30))
    return 0;
  return inSize + (16 + 4 + 1 + 4 + 1 + 64) * 2 + (3 * 4) + 1;
}

/* rle_decompress_additional_size: rle8_extreme_cpu.c:17-20 */
uint32_t hso_decompress_additional_size(void) { return 128; }

/* ------------------------------------------------------------------------------------------ */
/* run discovery (SURVEY.md A.3 "Run discovery common to all multi-symbol encoders")          */
/*   8 bit : maximal runs of equal bytes of length >= 2   (rle8_extreme_cpu.h:1062-1089)       */
/*   S > 1 : first p with d[p..p+S) == d[p+S..p+2S), extended by whole symbols and -- for the */
/*           byte-aligned variants -- the matching leading bytes of the next partial symbol   */
/*           (rleX_extreme_cpu_encode.h:79-163, :315-371).  Bytes >= n never match.           */

typedef struct {
  const uint8_t *d;
  uint32_t n;
  int S;
  int aligned;
  uint64_t i;
} runs_t;

static int runs_next(runs_t *r, uint32_t *pStart, uint32_t *pEnd)
{
  const uint8_t *d = r->d;
  const uint64_t n = r->n;
  const uint64_t S = (uint64_t)r->S;
  uint64_t p = r->i;

  if (S == 1)
  {
    while (p + 1 < n)
    {
      if (d[p] == d[p + 1])
      {
        uint64_t e = p + 2;
        while (e < n && d[e] == d[p])
          e++;
        *pStart = (uint32_t)p;
        *pEnd = (uint32_t)e;
        r->i = e;
        return 1;
      }
      p++;
    }
    r->i = n;
    return 0;
  }

  for (; p + 2 * S <= n; p++)
    if (memcmp(d + p, d + p + S, S) == 0)
      break;

  if (p + 2 * S > n)
  {
    r->i = n;
    return 0;
  }

  uint64_t e = p + 2 * S;

  while (e + S <= n && memcmp(d + e, d + p, S) == 0)
    e += S;

  if (!r->aligned && e + S <= n)
  {
    uint64_t j = 0;
    while (j < S && d[e + j] == d[p + j])
      j++;
    e += j;
  }

  *pStart = (uint32_t)p;
  *pEnd = (uint32_t)e;
  r->i = e;
  return 1;
}

/* ------------------------------------------------------------------------------------------ */
/* 8 bit multi, plain + Packed (canonical = AVX2 body)                                        */
/*   rle8_extreme_cpu.h:86-344 (wrapper, scalar tail, final block), :936-1099 (avx2 body)      */
/*   thresholds rle8_extreme_cpu.h:5-6, :15-17;  body/tail closed form SURVEY.md A.5 q1       */

static uint32_t enc8_multi(const uint8_t *d, uint32_t n, int packed, uint8_t *out)
{
  sink_t s = { out, 0 };
  put32(&s, n);
  put32(&s, 0);
  put8(&s, 0); /* mode = multi */

  const uint32_t SHORT = packed ? 3 : 6;
  uint32_t lastRLE = 0;
  uint8_t last = 0;
  int ended = 0;
  runs_t r = { d, n, 1, 0, 0 };
  uint32_t p, e;

  while (runs_next(&r, &p, &e))
  {
    const uint8_t sym = d[p];
    const uint32_t count = e - p;
    const uint32_t range = p - lastRLE + 1;
    int emit, same = 0;

    if (!packed)
    {
      emit = count >= 6;
    }
    else
    {
      const int64_t k = ((int64_t)e - p - 1) / 32;
      const int body = (e < n) && ((int64_t)p + 1 + 32 * k < (int64_t)n - 32);

      if (body)
      {
        same = sym == last;
        emit = count >= 11 || (range <= 127 && ((same && count >= 3) || count >= 4));
        if (emit)
          last = sym;
      }
      else
      {
        emit = count >= 11;
      }
    }

    if (!emit)
      continue;

    const uint32_t c = count - SHORT + 1;

    if (!packed)
    {
      put8(&s, sym);
      if (c <= 255) put8(&s, c); else { put8(&s, 0); put32(&s, c); }
      if (range <= 255) put8(&s, range); else { put8(&s, 0); put32(&s, range); }
    }
    else
    {
      const uint32_t sm = same ? 0x80 : 0;
      if (c <= 127) put8(&s, c | sm); else { put8(&s, sm); put32(&s, c); }
      if (!same) put8(&s, sym);
      if (range <= 127) put8(&s, range << 1); else put32(&s, (range << 1) | 1);
    }

    putn(&s, d + lastRLE, p - lastRLE);
    lastRLE = e;

    if (e >= n)
    {
      if (!packed) { put8(&s, 0); put8(&s, 0); put32(&s, 0); put8(&s, 0); put32(&s, 0); }
      else { put8(&s, 0x80); put32(&s, 0); put32(&s, 1); }
      ended = 1;
    }
  }

  if (!ended)
  {
    const uint32_t k = n - lastRLE;
    if (!packed) { put8(&s, 0); put8(&s, 0); put32(&s, 0); put8(&s, 0); put32(&s, k + 1); }
    else { put8(&s, 0x80); put32(&s, 0); put32(&s, ((k + 1) << 1) | 1); }
    putn(&s, d + lastRLE, k);
  }

  patch32(out, 4, (uint32_t)s.at);
  return (uint32_t)s.at;
}

/* ------------------------------------------------------------------------------------------ */
/* S-byte plain / Packed, S in {2,3,4,6,8}                                                    */
/*   rleX_extreme_cpu_encode.h:14-609 (16/32/64), rle24_extreme_cpu_encode.h, rle48_..._encode.h */
/*   thresholds rleX_extreme_cpu.h:1-16; sym-aligned Packed hybrid: SURVEY.md A.5 q10          */

typedef struct {
  int S, aligned, packed, range7;
  uint32_t SHORT, MEDIUM, LONG, MAXRANGE;
  uint8_t last[16];
} gparams_t;

static void gparams_init(gparams_t *g, int S, int aligned, int packed)
{
  g->S = S;
  g->aligned = aligned;
  g->packed = packed;
  g->range7 = packed && !aligned;
  memset(g->last, 0, sizeof(g->last));

  if (!packed)
  {
    g->SHORT = (uint32_t)S + 4;
    g->MEDIUM = 0;
    g->LONG = (uint32_t)S + 11;
    g->MAXRANGE = 255;
  }
  else
  {
    g->SHORT = 3;
    g->MEDIUM = (uint32_t)S + 3;
    g->LONG = g->range7 ? (uint32_t)S + 11 : (uint32_t)S + 10;
    g->MAXRANGE = g->range7 ? 127 : 255;
  }
}

/* 0 = keep as literals, 1 = short-range packet, 2 = long-range packet
 * (rleX_extreme_cpu_encode.h:174-311) */
static int g_decide(const gparams_t *g, const uint8_t *sym, uint32_t count, uint32_t range)
{
  int shortOk;

  if (!g->packed)
    shortOk = range <= g->MAXRANGE && count >= g->SHORT;
  else
    shortOk = range <= g->MAXRANGE && ((count >= g->SHORT && memcmp(sym, g->last, (size_t)g->S) == 0) || count >= g->MEDIUM);

  if (shortOk)
    return 1;
  if (count >= g->LONG)
    return 2;
  return 0;
}

static void g_put_run(gparams_t *g, sink_t *s, const uint8_t *sym, uint32_t count, uint32_t range, int longForm)
{
  const uint32_t S = (uint32_t)g->S;
  const uint32_t c = g->aligned ? (count / S - g->SHORT / S + 1) : (count - g->SHORT + 1);

  if (!g->packed)
  {
    putn(s, sym, S);
    if (c <= 255) put8(s, c); else { put8(s, 0); put32(s, c); }
  }
  else
  {
    const uint32_t same = memcmp(sym, g->last, S) == 0 ? 0x80 : 0;
    memcpy(g->last, sym, S);
    if (c <= 127) put8(s, c | same); else { put8(s, same); put32(s, c); }
    if (!same) putn(s, sym, S);
  }

  if (!longForm)
    put8(s, g->range7 ? (range << 1) & 0xFF : range);
  else if (g->range7)
    put32(s, (range << 1) | 1);
  else
  {
    put8(s, 0);
    put32(s, range);
  }
}

/* terminators: rleX_extreme_cpu_encode.h:384-603.  endRange7: the 128-bit encoder always
 * writes the plain `00, u32 0` end marker. */
static void g_put_term_end(const gparams_t *g, sink_t *s, int plainEndAlways)
{
  if (!g->packed) { putzeros(s, (size_t)g->S); put8(s, 0); put32(s, 0); }
  else { put8(s, 0x80); put32(s, 0); }

  if (g->range7 && !plainEndAlways) put32(s, 1);
  else { put8(s, 0); put32(s, 0); }
}

static void g_put_term_literals(const gparams_t *g, sink_t *s, const uint8_t *lit, uint32_t k)
{
  if (!g->packed) { putzeros(s, (size_t)g->S); put8(s, 0); put32(s, 0); }
  else { put8(s, 0x80); put32(s, 0); }

  if (g->range7) put32(s, ((k + 1) << 1) | 1);
  else { put8(s, 0); put32(s, k + 1); }

  putn(s, lit, k);
}

static uint32_t enc_generic(const uint8_t *d, uint32_t n, int S, int aligned, int packed, uint8_t *out)
{
  sink_t s = { out, 0 };
  put32(&s, n);
  put32(&s, 0);

  gparams_t g;
  gparams_init(&g, S, aligned, packed);

  uint32_t lastRLE = 0;
  int ended = 0;
  runs_t r = { d, n, S, aligned, 0 };
  uint32_t p, e;

  while (runs_next(&r, &p, &e))
  {
    const uint32_t count = e - p;
    const uint32_t range = p - lastRLE + 1;
    const int k = g_decide(&g, d + p, count, range);

    if (!k)
      continue;

    g_put_run(&g, &s, d + p, count, range, k == 2);
    putn(&s, d + lastRLE, p - lastRLE);
    lastRLE = e;

    if (e >= n)
    {
      g_put_term_end(&g, &s, 0);
      ended = 1;
    }
  }

  if (!ended)
    g_put_term_literals(&g, &s, d + lastRLE, n - lastRLE);

  patch32(out, 4, (uint32_t)s.at);
  return (uint32_t)s.at;
}

/* ------------------------------------------------------------------------------------------ */
/* 128 bit (S = 16): literal restatement of rle128_extreme_cpu.h:32-497       */

static void load16z(const uint8_t *d, int64_t n, int64_t i, uint8_t *dst)
{
  memset(dst, 0, 16);
  if (i < n)
    memcpy(dst, d + i, (size_t)((n - i) < 16 ? (n - i) : 16));
}

static uint32_t enc128(const uint8_t *d, uint32_t n32, int aligned, int packed, uint8_t *out)
{
  sink_t s = { out, 0 };
  put32(&s, n32);
  put32(&s, 0);

  gparams_t g;
  gparams_init(&g, 16, aligned, packed);

  const int64_t n = n32;
  int64_t i = 0, lastRLE = 0, count = 0;
  uint8_t symbol[16];
  load16z(d, n, 0, symbol); /* NOT inverted: q4 */

  while (i < n)
  {
    int restart = 1;

    while (restart)
    {
      restart = 0;

      /* extend the current run (also the very first step: the first block matches itself) */
      while (i < n - 16)
      {
        if (memcmp(d + i, symbol, 16) == 0)
        {
          count += 16;
          i += 16;
        }
        else
        {
          if (!aligned)
          {
            int64_t off = 0;
            while (d[i + off] == symbol[off])
              off++;
            i += off;
            count += off;
          }
          break;
        }
      }

      {
        const uint32_t range = (uint32_t)(i - lastRLE - count + 1);
        const int k = g_decide(&g, symbol, (uint32_t)count, range);

        if (k)
        {
          g_put_run(&g, &s, symbol, (uint32_t)count, range, k == 2);
          putn(&s, d + lastRLE, (size_t)(i - count - lastRLE));
          lastRLE = i;
        }
      }

      /* pair search, skipping past the highest mismatching byte */
      while (i < n - 32)
      {
        const uint8_t *a = d + i, *b = d + i + 16;

        if (memcmp(a, b, 16) == 0)
        {
          memcpy(symbol, a, 16);
          i += 32;
          count = 32;
          restart = 1;
          break;
        }
        else if (a[15] != b[15])
        {
          i += 16;
        }
        else
        {
          int hb = 14;
          while (a[hb] == b[hb])
            hb--;
          i += hb + 1;
        }
      }
    }

    /* scalar step; bytes >= n never match */
    load16z(d, n, i, symbol);

    if (i + 32 <= n && memcmp(d + i, d + i + 16, 16) == 0)
    {
      count = 32;
      i += 32;
    }
    else
    {
      count = 0;
      i += 1;
    }
  }

  {
    const uint32_t range = (uint32_t)(i - lastRLE - count + 1);
    const int k = g_decide(&g, symbol, (uint32_t)count, range);

    if (k)
    {
      g_put_run(&g, &s, symbol, (uint32_t)count, range, k == 2);
      putn(&s, d + lastRLE, (size_t)(i - count - lastRLE));
      g_put_term_end(&g, &s, 1);
    }
    else
    {
      g_put_term_literals(&g, &s, d + lastRLE, (uint32_t)(i - lastRLE));
    }
  }

  patch32(out, 4, (uint32_t)s.at);
  return (uint32_t)s.at;
}

/* ------------------------------------------------------------------------------------------ */
/* 3 / 7 symbol LUT:  rleX_Xsl.h:93-264 (state, process_symbol), :269-346 (8 bit wrapper),    */
/*   rleX_Xsl_multibyte_encoder.h:18-370 (S > 1).  Thresholds rleX_Xsl.h:1-17.                */

static uint32_t enc_lut(const uint8_t *d, uint32_t n, int S, int aligned, int K, uint8_t *out)
{
  sink_t s = { out, 0 };
  put32(&s, n);
  put32(&s, 0);

  const uint32_t RB = (K == 3) ? 7 : 6;
  const uint32_t MAXC = 127, MAXR = (1u << RB) - 1;
  static const uint8_t init[7] = { 0x00, 0x7F, 0xFF, 0x01, 0x7E, 0x80, 0xFE };
  uint8_t lut[7][16];

  for (int k = 0; k < K; k++)
    memset(lut[k], init[k], 16);

  uint32_t lastRLE = 0;
  int ended = 0;
  runs_t r = { d, n, S, (S == 1) ? 0 : aligned, 0 };
  uint32_t p, e;

  while (runs_next(&r, &p, &e))
  {
    const uint8_t *sym = d + p;
    const uint32_t count = e - p;
    const uint32_t range = p - lastRLE + 2;
    int m = K;

    for (int k = 0; k < K; k++)
      if (memcmp(lut[k], sym, (size_t)S) == 0) { m = k; break; }

    const uint32_t c = (!aligned || S == 1) ? (count - 3 + 2) : (count / (uint32_t)S - 3 / (uint32_t)S + 2);

    /* penalty uses 0xFFFFF where the writer uses 0xFFFF (q3, rleX_Xsl.h:130 vs :195) */
    uint32_t pen = (range <= 0xFFFFF) ? (range <= MAXR ? 0 : 2) : 4;
    pen += (c <= 0xFFFFF) ? (c <= MAXC ? 0 : 2) : 4;
    pen += (m == K) ? 1 : 0;

    if (!(count >= (uint32_t)S + 10 || count >= 3 + pen))
      continue;

    /* move to front (rleX_Xsl.h:134-188) */
    {
      uint8_t tmp[16];
      memcpy(tmp, sym, (size_t)S);
      const int from = (m == K) ? K - 1 : m;
      for (int k = from; k > 0; k--)
        memcpy(lut[k], lut[k - 1], 16);
      memset(lut[0], 0, 16);
      memcpy(lut[0], tmp, (size_t)S);
    }

    const uint32_t c7 = (c <= MAXC) ? c : (c <= 0xFFFF ? 1 : 0);
    const uint32_t r7 = (range <= MAXR) ? range : (range <= 0xFFFF ? 1 : 0);
    const uint32_t v = ((uint32_t)m << (K == 3 ? 14 : 13)) | (c7 << RB) | r7;

    put16(&s, v);
    if (m == K) putn(&s, sym, (size_t)S);
    if (c != c7) { if (c <= 0xFFFF) put16(&s, c); else put32(&s, c); }
    if (range != r7) { if (range <= 0xFFFF) put16(&s, range); else put32(&s, range); }

    putn(&s, d + lastRLE, p - lastRLE);
    lastRLE = e;

    if (e >= n)
    {
      put16(&s, (1u << RB) | 1);
      put16(&s, 0);
      put16(&s, 0);
      ended = 1;
    }
  }

  if (!ended)
  {
    const uint32_t k = n - lastRLE;
    put16(&s, 1u << RB);
    put16(&s, 0);
    put32(&s, k + 2);
    putn(&s, d + lastRLE, k);
  }

  patch32(out, 4, (uint32_t)s.at);
  return (uint32_t)s.at;
}

/* ------------------------------------------------------------------------------------------ */
/* Short family: 0 / 1 / 3 / 7 symbol LUT with one-byte packed headers.       */
/*   parameters       rleX_Xsl_short.h:1-43                                                    */
/*   process_symbol   rleX_Xsl_short.h:152-372  (decision, move to front, header, literals)    */
/*   8 bit wrapper    rleX_Xsl_short.h:380-523 (+ bodies :529-667: run discovery = maximal     */
/*                    runs of equal bytes, as in the other 8 bit multi encoders)               */
/*   S > 1            rleX_Xsl_short_multibyte_encoder.h:18-412 (same run discovery as         */
/*                    rleX_Xsl_multibyte_encoder.h)                                            */

typedef struct { uint32_t LB, CB, RBP, RB, CINV, MAXPR, MAXPC, MAXTR, MINS, MINL, TB; } short_params_t;

static void short_params(short_params_t *q, int S, int K)
{
  q->LB = (K == 3) ? 2 : (K == 1 ? 1 : (K == 0 ? 0 : 3));
  q->CB = (K == 3 || K == 1) ? 3 : (K == 0 ? 4 : 2);
  q->RBP = 8 - q->LB - q->CB;
  q->RB = (K != 7) ? 24 - q->LB - q->CB - 9 : 24 - q->LB - q->RBP - 9;
  q->CINV = (1u << q->CB) - 1;
  q->MAXPR = (1u << q->RBP) - 1;
  q->MAXPC = (1u << q->CB) - 2;
  q->MAXTR = (1u << q->RB) - 1;
  q->MINS = (K != 0) ? 2u : (uint32_t)S + 2;
  q->MINL = (K != 0) ? (uint32_t)S + 11 : (uint32_t)S + 12;
  q->TB = (K == 3) ? 4u : (K == 7 ? 2u : 8u);   /* second byte of the terminators (:474-482) */
}

typedef struct {
  sink_t s;
  short_params_t q;
  int S, K, al;
  int single;            /* rle8_single_short: one symbol for the whole stream, none in the packets */
  uint8_t lut[7][16];
  uint32_t lastRLE;
  const uint8_t *d;
} short_state_t;

static void short_init(short_state_t *st, const uint8_t *d, uint32_t n, int S, int aligned, int K, uint8_t *out)
{
  static const uint8_t init[7] = { 0x00, 0x7F, 0xFF, 0x01, 0x7E, 0x80, 0xFE };
  st->s.o = out; st->s.at = 0;
  put32(&st->s, n);
  put32(&st->s, 0);
  short_params(&st->q, S, K);
  st->S = S; st->K = K; st->al = (S == 1) ? 0 : aligned;
  st->single = 0;
  for (int k = 0; k < K; k++)
    memset(st->lut[k], init[k], 16);
  st->lastRLE = 0;
  st->d = d;
}

/* process_symbol (rleX_Xsl_short.h:152-372) for the run [i - count, i) of `sym`: decide, and if the run is stored write the packet */
static int short_process(short_state_t *st, const uint8_t *sym, int64_t count, uint32_t i)
{
  const short_params_t *q = &st->q;
  const int S = st->S, K = st->K;
  const int64_t range = (int64_t)i - st->lastRLE - count + 2;
  int m = K;

  for (int k = 0; k < K; k++)
    if (memcmp(st->lut[k], sym, (size_t)S) == 0) { m = k; break; }

  const int64_t sc = !st->al ? count - q->MINS + 2 : count / S - q->MINS / (uint32_t)S + 2;
  const uint64_t count3 = (uint64_t)(sc - 2), range3 = (uint64_t)(range - 2);
  const int single = range3 <= q->MAXPR && count3 <= q->MAXPC;
  const int is19 = sc <= 511 && range <= (int64_t)q->MAXTR;
  int64_t pen = (K > 0 && m == K) ? S : 0;

  if (!single)
  {
    pen += 2;
    if (!is19)
      pen += (range <= 0xFFFFF ? (range <= (int64_t)q->MAXTR ? 0 : 2) : 4) + (sc <= 0xFFFFF ? (sc <= 511 ? 0 : 2) : 4);
  }

  if (!(count >= (int64_t)q->MINL || count >= (int64_t)q->MINS + pen))
    return 0;

  if (K > 0)
  {
    uint8_t tmp[16];
    memset(tmp, 0, 16);
    memcpy(tmp, sym, (size_t)S);
    const int from = (m == K) ? K - 1 : m;
    for (int k = from; k > 0; k--)
      memcpy(st->lut[k], st->lut[k - 1], 16);
    memcpy(st->lut[0], tmp, 16);
  }

  const uint32_t mi = (K > 0) ? (uint32_t)m : 0;
  sink_t *s = &st->s;

  if (single)
    put8(s, (mi << (q->CB + q->RBP)) | ((uint32_t)count3 << q->RBP) | (uint32_t)range3);
  else
  {
    const uint32_t scx = (sc <= 511) ? (uint32_t)sc : (sc <= 0xFFFF ? 1u : 0u);
    const uint32_t rx = (range <= (int64_t)q->MAXTR) ? (uint32_t)range : (range <= 0xFFFF ? 1u : 0u);
    put8(s, (mi << (q->CB + q->RBP)) | (q->CINV << q->RBP) | (((scx << (q->RB - 8)) >> 8) & 0xFF));
    put8(s, ((scx << (q->RB - 8)) | (rx >> 8)) & 0xFF);
    put8(s, rx & 0xFF);
    if ((int64_t)scx != sc) { if (sc <= 0xFFFF) put16(s, (uint32_t)sc); else put32(s, (uint32_t)sc); }
    if ((int64_t)rx != range) { if (range <= 0xFFFF) put16(s, (uint32_t)range); else put32(s, (uint32_t)range); }
  }

  if (!st->single && (K == 0 || m == K))
    putn(s, sym, (size_t)S);

  const uint32_t p = (uint32_t)((int64_t)i - count);
  putn(s, st->d + st->lastRLE, p - st->lastRLE);
  st->lastRLE = i;
  return 1;
}

static uint32_t short_finish(short_state_t *st, uint32_t n, int lastStored, uint8_t *out)
{
  const short_params_t *q = &st->q;
  sink_t *s = &st->s;

  if (lastStored)
  {
    put8(s, q->CINV << q->RBP); put8(s, q->TB); put8(s, 1); put16(s, 0); put16(s, 0);
    if (st->K == 0 && !st->single) put8(s, 0);    /* one byte, whatever the symbol width (:497-500, multibyte :370-373) */
  }
  else
  {
    const uint32_t k = n - st->lastRLE;
    put8(s, q->CINV << q->RBP); put8(s, q->TB); put8(s, 0); put16(s, 0); put32(s, k + 2);
    if (st->K == 0 && !st->single) putzeros(s, (size_t)st->S);   /* a whole zero symbol here (:517-520, multibyte :398-401) */
    putn(s, st->d + st->lastRLE, k);
  }

  patch32(out, 4, (uint32_t)s->at);
  return (uint32_t)s->at;
}

static uint32_t enc_short(const uint8_t *d, uint32_t n, int S, int aligned, int K, uint8_t *out)
{
  short_state_t st;
  short_init(&st, d, n, S, aligned, K, out);
  runs_t r = { d, n, S, st.al, 0 };
  uint32_t p, e;
  int ended = 0;

  while (runs_next(&r, &p, &e))
    if (short_process(&st, d + p, (int64_t)e - p, e) && e >= n)
      ended = 1;

  return short_finish(&st, n, ended, out);
}

/* Greedy encoders of the byte-aligned 1/3/7 symbol LUT Short codecs: rleX_Xsl_short.h:746-1000, restated step for step.    */
/* They also try the leading bytes of the listed symbols as (partial) runs; the streams decode with the Short decoders.       */
/* Bytes at or beyond n never match (the reference reads the symbol behind a fitting one without a bound, :884-888).           */
static uint32_t common_prefix(const uint8_t *a, const uint8_t *b, int S) { int j = 0; while (j < S && a[j] == b[j]) j++; return (uint32_t)j; }

static uint32_t enc_short_greedy(const uint8_t *d, uint32_t n, int S, int K, uint8_t *out)
{
  short_state_t st;
  short_init(&st, d, n, S, 0, K, out);
  uint8_t sym[16];
  memset(sym, 0, 16);
  for (int j = 0; j < S; j++) sym[j] = (uint8_t)~((uint32_t)j < n ? d[j] : 0);
  int64_t count = 0;
  uint64_t i = 0;

  while (i < n)
  {
    if (count && i + (uint64_t)S <= n)
    {
      if (memcmp(d + i, sym, (size_t)S) == 0) { count += S; i += (uint64_t)S; continue; }
      if (S == 2) { if (sym[0] == d[i]) { count++; i++; } }
      else { const uint32_t j = common_prefix(sym, d + i, S); i += j; count += j; }
    }

    for (;;)   /* label not_a_full_match_but_a_match (:861) */
    {
      short_process(&st, sym, count, (uint32_t)i);

      memset(sym, 0, 16);
      for (int j = 0; j < S; j++) sym[j] = (i + (uint64_t)j < n) ? d[i + j] : 0;
      const int fits = i + (uint64_t)S <= n;

      if (fits && i + 2 * (uint64_t)S <= n && memcmp(d + i + S, sym, (size_t)S) == 0) { count = 2 * S; i += 2 * (uint64_t)S; break; }

      if (!fits) { count = 0; i++; break; }

      uint32_t pc = 0;
      int idx = 0;

      for (int j = 0; j < K; j++)
      {
        const uint32_t c = common_prefix(st.lut[j], sym, S);
        if (c == (uint32_t)S) { idx = j; pc = (uint32_t)S; break; }
        if (S != 2 && c > pc) { idx = j; pc = c; }
      }

      if (S != 2 ? pc >= st.q.MINS : pc != 0)
      {
        count = pc; i += pc;
        memcpy(sym, st.lut[idx], 16);
        if (S != 2 && count < S) continue;      /* goto not_a_full_match_but_a_match */
        break;
      }

      count = 0; i++;
      break;
    }
  }

  return short_finish(&st, n, short_process(&st, sym, count, (uint32_t)i), out);
}

/* ------------------------------------------------------------------------------------------ */
/* 8 bit Single: literal restatement                                          */
/*   symbol pick  rle8_extreme_cpu.c:53-153 (the sse2 estimator is used on every ISA, q9)      */
/*   encoder      rle8_extreme_cpu.h:346-700, :1103-1321                                       */

static int all16(const uint8_t *p, uint8_t v) { for (int k = 0; k < 16; k++) if (p[k] != v) return 0; return 1; }
static int any16(const uint8_t *p, uint8_t v) { for (int k = 0; k < 16; k++) if (p[k] == v) return 1; return 0; }

static uint8_t single_pick_symbol(const uint8_t *d, uint32_t n32)
{
  uint32_t probL[256], pcL[256];
  memset(probL, 0, sizeof(probL));
  memset(pcL, 0, sizeof(pcL));

  const int64_t n = n32;
  if (d[0] != 0)
    pcL[0] = 0xFFFFFFFFu;

  int64_t i = 0;
  const int64_t end = n - 16;
  uint8_t last = (uint8_t)~d[0];
  uint32_t count = 0;

  while (i < end)
  {
    if (all16(d + i, last))
    {
      count += 15; /* sic */
      i += 15;
    }
    else
    {
      if (any16(d + i, last) || count > 1)
      {
        int z = 0;
        while (d[i + z] == last)
          z++;
        count += (uint32_t)z;
        i += z;
        probL[last] += count;
        pcL[last]++;
      }

      while (i < end)
      {
        int f = -1;
        for (int k = 0; k < 15; k++)
          if (d[i + k] == d[i + k + 1]) { f = k; break; }

        if (f < 0)
          i += 15;
        else
        {
          i += f;
          break;
        }
      }

      count = 1;
      last = d[i];
    }

    i++;
  }

  probL[last] += count;
  pcL[last]++;

  uint32_t best = 0;
  uint8_t bestSym = 0;

  for (int sy = 0; sy < 256; sy++)
  {
    if (pcL[sy] > 0 && probL[sy] / pcL[sy] > 2)
    {
      const uint32_t saved = probL[sy] - pcL[sy] * 2;
      if (saved > best)
      {
        best = saved;
        bestSym = (uint8_t)sy;
      }
    }
  }

  return bestSym;
}

static void single_put_count(sink_t *s, uint32_t count, uint32_t SHORT)
{
  const uint32_t c = count - SHORT + 1;
  if (c <= 255) put8(s, c); else { put8(s, 0); put32(s, c); }
}

static uint32_t enc8_single(const uint8_t *d, uint32_t n32, int packed, uint8_t *out)
{
  const uint32_t SHORT = packed ? 2 : 4;
  const uint32_t MEDIUM = 6; /* Packed only */
  const uint32_t LONG = packed ? 10 : 8;
  const uint8_t sym = single_pick_symbol(d, n32);

  sink_t s = { out, 0 };
  put32(&s, n32);
  put32(&s, 0);
  put8(&s, 1); /* mode = single */
  put8(&s, sym);

  const int64_t n = n32;
  const int64_t end = n - 16;
  int64_t i = 0, count = 0, lastRLE = 0, wasted = 0, firstW = 0;

  while (i < end)
  {
    if (all16(d + i, sym))
    {
      count += 16;
      i += 15;
    }
    else
    {
      if (any16(d + i, sym) || count > 1)
      {
        int z = 0;
        while (d[i + z] == sym)
          z++;
        count += z;
        i += z;

        const int64_t range = i - lastRLE - count + 1;

        if (count >= SHORT)
        {
          if (range <= 255)
          {
            single_put_count(&s, (uint32_t)count, SHORT);
            put8(&s, (uint32_t)range);
            putn(&s, d + lastRLE, (size_t)(i - count - lastRLE));
            lastRLE = i;
            wasted = 0;
          }
          else if (count >= LONG || (packed && (count - SHORT + 1 <= 255 && count >= MEDIUM)))
          {
            single_put_count(&s, (uint32_t)count, SHORT);
            put8(&s, 0);
            put32(&s, (uint32_t)range);
            putn(&s, d + lastRLE, (size_t)(i - count - lastRLE));
            lastRLE = i;
            wasted = 0;
          }
          else
          {
            wasted++;

            if (wasted == 1 || i - firstW > 255)
            {
              firstW = i - count;
              wasted = 1;
            }
            else if (wasted > 2)
            {
              /* back-track to the first skipped run and force a long packet */
              i = firstW;
              wasted = 0;
              count = 0;

              while (i < end && d[i] == sym)
              {
                count++;
                i++;
              }

              put8(&s, (uint32_t)(count - SHORT + 1) & 0xFF);
              put8(&s, 0);
              put32(&s, (uint32_t)(i - lastRLE - count + 1));
              putn(&s, d + lastRLE, (size_t)(i - count - lastRLE));
              lastRLE = i;
            }
          }
        }
      }

      count = 0;

      while (i < end)
      {
        int pop = 0, first = -1;
        for (int k = 0; k < 16; k++)
          if (d[i + k] == sym) { pop++; if (first < 0) first = k; }

        if (pop == 0 || (d[i + 15] != sym && (uint32_t)pop < SHORT))
          i += 16;
        else
        {
          i += first;
          count = 1;
          break;
        }
      }
    }

    i++;
  }

  for (; i < n; i++)
  {
    if (d[i] == sym)
      count++;
    else
    {
      const int64_t range = i - lastRLE - count + 1;

      if (range <= 255 && count >= SHORT)
      {
        single_put_count(&s, (uint32_t)count, SHORT);
        put8(&s, (uint32_t)range);
        putn(&s, d + lastRLE, (size_t)(i - count - lastRLE));
        lastRLE = i;
      }
      else if (count >= LONG)
      {
        single_put_count(&s, (uint32_t)count, SHORT);
        put8(&s, 0);
        put32(&s, (uint32_t)range);
        putn(&s, d + lastRLE, (size_t)(i - count - lastRLE));
        lastRLE = i;
      }

      count = 0;
    }
  }

  {
    const int64_t range = i - lastRLE - count + 1;

    if (range <= 255 && count >= SHORT)
    {
      single_put_count(&s, (uint32_t)count, SHORT);
      put8(&s, (uint32_t)range);
      putn(&s, d + lastRLE, (size_t)(i - count - lastRLE));
      put8(&s, 0); put32(&s, 0); put8(&s, 0); put32(&s, 0);
    }
    else if (count >= LONG)
    {
      single_put_count(&s, (uint32_t)count, SHORT);
      put8(&s, 0);
      put32(&s, (uint32_t)range);
      putn(&s, d + lastRLE, (size_t)(i - count - lastRLE));
      put8(&s, 0); put32(&s, 0); put8(&s, 0); put32(&s, 0);
    }
    else
    {
      put8(&s, 0); put32(&s, 0); put8(&s, 0); put32(&s, (uint32_t)(range + count));
      putn(&s, d + lastRLE, (size_t)(i - lastRLE));
    }
  }

  patch32(out, 4, (uint32_t)s.at);
  return (uint32_t)s.at;
}

/* ------------------------------------------------------------------------------------------ */
/* decoders (SURVEY.md A.1 packet grammar)                                                    */
/*   plain/Packed: rleX_extreme_cpu_decode.h:27-164, rle8_extreme_cpu.h:1546-2434,            */
/*                 rle24/48_extreme_cpu_decode.h, rle128_extreme_cpu.h:499-776                 */
/*   LUT:          rleX_Xsl.h:530-1844                                                         */
/* Unlike the reference these never write outside [0, U) and never read outside the stream;  */
/* a malformed stream returns 0.                                                              */

typedef struct {
  const uint8_t *b;
  size_t p, n;
  int bad;
} src_t;

static uint32_t rd8(src_t *s) { if (s->p + 1 > s->n) { s->bad = 1; return 0; } return s->b[s->p++]; }
static uint32_t rd16(src_t *s) { if (s->p + 2 > s->n) { s->bad = 1; return 0; } uint32_t v = get16(s->b + s->p); s->p += 2; return v; }
static uint32_t rd32(src_t *s) { if (s->p + 4 > s->n) { s->bad = 1; return 0; } uint32_t v = get32(s->b + s->p); s->p += 4; return v; }
static void rdn(src_t *s, uint8_t *dst, size_t k) { if (s->p + k > s->n) { s->bad = 1; memset(dst, 0, k); return; } memcpy(dst, s->b + s->p, k); s->p += k; }

typedef struct {
  uint8_t *o;
  uint64_t at, cap;
  int bad;
} dst_t;

static void out_copy(dst_t *o, src_t *s, uint64_t k)
{
  if (s->p + k > s->n || o->at + k > o->cap) { s->bad = 1; return; }
  memcpy(o->o + o->at, s->b + s->p, (size_t)k);
  s->p += (size_t)k;
  o->at += k;
}

static void out_fill(dst_t *o, const uint8_t *sym, uint32_t S, uint64_t k)
{
  if (o->at + k > o->cap) { o->bad = 1; return; }
  uint8_t *q = o->o + o->at;
  if (S == 1)
    memset(q, sym[0], (size_t)k);
  else
    for (uint64_t j = 0; j < k; j++)
      q[j] = sym[j % S];
  o->at += k;
}

/* plain / Packed / Single.  `shortv` = SHORT threshold of the codec. */
static int dec_packets(src_t *s, dst_t *o, int S, int aligned, int packed, int range7, int single, uint32_t shortv, uint8_t *sym)
{
  for (;;)
  {
    uint32_t cnt, range;

    if (single)
    {
      cnt = rd8(s);
      if (cnt == 0) cnt = rd32(s);
      range = rd8(s);
      if (range == 0) { range = rd32(s); if (range == 0) break; }
    }
    else if (!packed)
    {
      rdn(s, sym, (size_t)S);
      cnt = rd8(s);
      if (cnt == 0) cnt = rd32(s);
      range = rd8(s);
      if (range == 0) { range = rd32(s); if (range == 0) break; }
    }
    else
    {
      const uint32_t x = rd8(s);
      cnt = x & 0x7F;
      if (cnt == 0) cnt = rd32(s);
      if (!(x & 0x80)) rdn(s, sym, (size_t)S);

      if (range7)
      {
        if (s->p < s->n && (s->b[s->p] & 1)) { range = rd32(s) >> 1; if (range == 0) break; }
        else range = rd8(s) >> 1;
      }
      else
      {
        range = rd8(s);
        if (range == 0) { range = rd32(s); if (range == 0) break; }
      }
    }

    if (s->bad)
      return 0;

    if (range > 1) /* a 7 bit range byte of 0x00 means "no literals" (q11) */
      out_copy(o, s, (uint64_t)range - 1);

    if (cnt == 0)
      break;

    const uint64_t runBytes = aligned ? ((uint64_t)cnt + shortv / (uint32_t)S - 1) * (uint32_t)S : (uint64_t)cnt + shortv - 1;
    out_fill(o, sym, (uint32_t)S, runBytes);

    if (s->bad || o->bad)
      return 0;
  }

  return !(s->bad || o->bad);
}

static int dec_lut(src_t *s, dst_t *o, int S, int aligned, int K)
{
  const uint32_t RB = (K == 3) ? 7 : 6;
  static const uint8_t init[7] = { 0x00, 0x7F, 0xFF, 0x01, 0x7E, 0x80, 0xFE };
  uint8_t lut[7][16];

  for (int k = 0; k < K; k++)
    memset(lut[k], init[k], 16);

  for (;;)
  {
    const uint32_t v = rd16(s);
    const uint32_t idx = v >> (K == 3 ? 14 : 13);
    uint32_t cnt = (v >> RB) & 0x7F;
    uint32_t range = v & ((1u << RB) - 1);

    if ((int)idx == K)
    {
      uint8_t t[16];
      memset(t, 0, 16);
      rdn(s, t, (size_t)S);
      for (int k = K - 1; k > 0; k--) memcpy(lut[k], lut[k - 1], 16);
      memcpy(lut[0], t, 16);
    }
    else if (idx > 0)
    {
      uint8_t t[16];
      memcpy(t, lut[idx], 16);
      for (int k = (int)idx; k > 0; k--) memcpy(lut[k], lut[k - 1], 16);
      memcpy(lut[0], t, 16);
    }

    if (cnt == 0) cnt = rd32(s);
    else if (cnt == 1) cnt = rd16(s);

    if (range == 0) range = rd32(s);
    else if (range == 1) { range = rd16(s); if (range == 0) break; }

    if (s->bad || range < 2)
      return 0;

    out_copy(o, s, (uint64_t)range - 2);

    if (cnt == 0)
      break;

    const uint64_t runBytes = (aligned && S > 1) ? ((uint64_t)cnt + 3 / (uint32_t)S - 2) * (uint32_t)S : (uint64_t)cnt + 3 - 2;
    out_fill(o, lut[0], (uint32_t)S, runBytes);

    if (s->bad || o->bad)
      return 0;
  }

  return !(s->bad || o->bad);
}

/* rle8_single_short (rleX_Xsl_short.h with SINGLE: wrapper :380-523, body compress_single_sse2 :1058-1120), restated step   */
/* for step: the body's skip loop passes over 16-byte windows with fewer than two occurrences of the symbol (unless the last   */
/* byte is one), and the position it stops at when the windows run out is never examined (the for loop's own i++).            */
static uint32_t enc_single_short(const uint8_t *d, uint32_t n, uint8_t *out)
{
  short_state_t st;
  short_init(&st, d, n, 1, 0, 0, out);
  st.single = 1;
  st.q.MINS = 2; st.q.MINL = 3 + 4 + 4;            /* :2-7 */
  const uint8_t symbol = single_pick_symbol(d, n);
  put8(&st.s, symbol);

  int64_t count = 0;
  int64_t i = 0;
  const int64_t end = (int64_t)n - 16;

  for (; i < end; i++)
  {
    uint32_t mask = 0;
    for (int k = 0; k < 16; k++) mask |= (uint32_t)(d[i + k] == symbol) << k;

    if (mask == 0xFFFF)
    {
      count += 16;
      i += 15;
      continue;
    }

    if (mask != 0 || count > 1)
    {
      uint32_t z = 0;
      while ((mask >> z) & 1) z++;
      count += z;
      i += z;
      short_process(&st, &symbol, count, (uint32_t)i);
    }

    count = 0;

    while (i < end)
    {
      uint32_t cmp = 0, pop = 0;
      for (int k = 0; k < 16; k++) if (d[i + k] == symbol) { cmp |= 1u << k; pop++; }

      if (cmp == 0 || ((cmp & 0x8000) == 0 && pop < 2))
        i += 16;
      else
      {
        uint32_t z = 0;
        while (!((cmp >> z) & 1)) z++;
        i += z;
        count = 1;
        break;
      }
    }
  }

  for (; i < (int64_t)n; i++)
  {
    if (d[i] == symbol)
      count++;
    else
    {
      short_process(&st, &symbol, count, (uint32_t)i);
      count = 0;
    }
  }

  return short_finish(&st, n, short_process(&st, &symbol, count, (uint32_t)i), out);
}

/* Short family decoder: rleX_Xsl_short.h:1207-1480 (sse body; every ISA body decodes the same grammar) */
static int dec_short(src_t *s, dst_t *o, int S, int aligned, int K, int single)
{
  short_params_t q;
  short_params(&q, S, K);
  static const uint8_t init[7] = { 0x00, 0x7F, 0xFF, 0x01, 0x7E, 0x80, 0xFE };
  uint8_t lut[7][16];
  uint8_t cur[16];

  memset(cur, 0, 16);                              /* K == 0: symbol starts as 0; K == 1: lut[0] = 0 */
  if (single) { q.MINS = 2; cur[0] = (uint8_t)rd8(s); }   /* rle8_single_short: the symbol sits behind the stream header (:1211-1216) */
  memset(lut[0], 0, 16);
  for (int k = 1; k < K; k++)
    memset(lut[k], init[k], 16);

  for (;;)
  {
    const uint32_t p1 = rd8(s);
    const uint32_t idx = (K > 0) ? p1 >> (q.CB + q.RBP) : 0;
    const uint32_t c3 = (p1 >> q.RBP) & q.CINV;
    uint32_t cnt, range;

    if (c3 == q.CINV)
    {
      const uint32_t p2 = rd8(s), p3 = rd8(s);
      cnt = (p2 >> (q.RB - 8)) | ((p1 & q.MAXPR) << (8 - (q.RB - 8)));
      range = p3 | ((p2 & ((1u << (q.RB - 8)) - 1)) << 8);

      if (cnt == 0) cnt = rd32(s);
      else if (cnt == 1) cnt = rd16(s);

      if (range == 0) range = rd32(s);
      else if (range == 1) { range = rd16(s); if (range == 0) break; }
    }
    else
    {
      cnt = c3 + 2;
      range = (p1 & q.MAXPR) + 2;
    }

    if (K > 0)
    {
      if ((int)idx == K)
      {
        uint8_t t[16];
        memset(t, 0, 16);
        rdn(s, t, (size_t)S);
        for (int k = K - 1; k > 0; k--) memcpy(lut[k], lut[k - 1], 16);
        memcpy(lut[0], t, 16);
      }
      else if (idx > 0)
      {
        if ((int)idx > K) return 0;
        uint8_t t[16];
        memcpy(t, lut[idx], 16);
        for (int k = (int)idx; k > 0; k--) memcpy(lut[k], lut[k - 1], 16);
        memcpy(lut[0], t, 16);
      }
      memcpy(cur, lut[0], 16);
    }
    else if (!single)
    {
      memset(cur, 0, 16);
      rdn(s, cur, (size_t)S);
    }

    if (s->bad || range < 2)
      return 0;

    out_copy(o, s, (uint64_t)range - 2);

    if (cnt == 0)
      break;

    const uint64_t runBytes = (aligned && S > 1) ? ((uint64_t)cnt + q.MINS / (uint32_t)S - 2) * (uint32_t)S : (uint64_t)cnt + q.MINS - 2;
    out_fill(o, cur, (uint32_t)S, runBytes);

    if (s->bad || o->bad)
      return 0;
  }

  return !(s->bad || o->bad);
}

/* ------------------------------------------------------------------------------------------ */
/* rle8m: the low-entropy codec in its sub-sectioned form, the format of the reference's GPU    */
/* (OpenCL) decoder.  src/rle8_low_entropy_cpu.c:                        */
/*   compress info  :254-338   (run statistics over the whole input, symbols by run count)     */
/*   info bytes     :441-472 / :545-606                                                         */
/*   section stream :474-543   (symbol, and behind a flagged symbol the code of how many more   */
/*                              of it follow, at most 254)                                      */
/*   container      :126-250   (u32 size, u32 inSize, u32 sections, u32 end offsets, info, ...)  */
/*   decoder        :930-1021, src/rle8_ocl_kernel.h:8-43 (every ISA variant decodes the same   */
/*                              grammar)                                                        */

uint32_t hso_rle8m_compress_bounds(uint32_t subSections, uint32_t inSize)
{
  return inSize + (256 / 8) + 1 + 256 + 4 * (2 + subSections - 1 + 1);
}

typedef struct { uint8_t rle[256]; uint8_t symbolsByProb[256]; uint8_t symbolCount; } le_info_t;

/* maxLen: 255, or 32 for the Short form (ULTRA_MAX_BLOCK_LENGTH, src/rle8_low_entropy_short_cpu.c:9, :571).  onlyMax: the rule of the
 * *_only_max_frequency encoders (src/rle8_low_entropy_cpu.c:340-439, src/rle8_low_entropy_short_cpu.c:622-720): one symbol -- the first
 * with the most bytes saved among those whose runs average more than 2 -- carries repeat codes, if any does. */
static void le_get_info_ex(const uint8_t *d, uint32_t n, le_info_t *info, uint32_t maxLen, int onlyMax)
{
  uint32_t prob[256], pcount[256];
  uint8_t consumed[256];
  memset(prob, 0, sizeof(prob));
  memset(pcount, 0, sizeof(pcount));
  memset(consumed, 0, sizeof(consumed));

  uint8_t last = 0;
  uint32_t count = 0;

  if (d[0] != last)
    pcount[last] = 0xFFFFFFFFu;

  for (uint32_t i = 0; i < n; i++)
  {
    if (d[i] == last)
      count++;
    else
    {
      prob[last] += count;
      pcount[last] += (count / maxLen) + 1;
      count = 1;
      last = d[i];
    }
  }

  prob[last] += count;
  pcount[last]++;

  if (!onlyMax)
  {
    for (int i = 0; i < 256; i++)
      info->rle[i] = (pcount[i] > 0) ? ((prob[i] / pcount[i]) >= 2) : 0;
  }
  else
  {
    size_t maxSaved = 0, maxAt = 0;
    memset(info->rle, 0, sizeof(info->rle));
    for (size_t i = 0; i < 256; i++)
      if (pcount[i] > 0 && prob[i] / pcount[i] > 2)
      {
        const size_t saved = prob[i] - (pcount[i] * 2);       /* (uint32 arithmetic, as in the reference: :392) */
        if (saved > maxSaved) { maxSaved = saved; maxAt = i; }
      }
    if (maxSaved > 0)
      info->rle[maxAt] = 1;
  }

  uint32_t remaining = 256;

  for (int i = 255; i >= 0; i--)
    if (pcount[i] == 0)
    {
      consumed[i] = 1;
      remaining--;
      info->symbolsByProb[remaining] = (uint8_t)i;
    }

  for (uint32_t index = 0; index < remaining; index++)
  {
    uint32_t max = 0;
    int maxIndex = 0;

    for (int i = 0; i < 256; i++)
      if (!consumed[i] && pcount[i] > max) { max = pcount[i]; maxIndex = i; }

    info->symbolsByProb[index] = (uint8_t)maxIndex;
    consumed[maxIndex] = 1;
  }

  info->symbolCount = (uint8_t)remaining;
}

static void le_get_info(const uint8_t *d, uint32_t n, le_info_t *info) { le_get_info_ex(d, n, info, 255, 0); }

static uint32_t le_write_info(const le_info_t *info, uint8_t *out)
{
  uint32_t index = 0;

  for (int i = 0; i < 32; i++)
  {
    uint8_t v = 0;
    for (int j = 0; j < 8; j++) v |= (uint8_t)((info->rle[j + i * 8] ? 1 : 0) << j);
    out[index++] = v;
  }

  out[index++] = info->symbolCount;
  const uint32_t sc = info->symbolCount ? info->symbolCount : 255;   /* sic: 256 symbols are stored as 0 and written as 255 (:460-464) */
  memcpy(out + index, info->symbolsByProb, sc);
  return index + sc;
}

static uint32_t le_compress_section_ex(const uint8_t *d, uint32_t n, const le_info_t *info, uint8_t *out, uint32_t maxLen)
{
  uint32_t index = 0;

  for (uint32_t i = 0; i < n; i++)
  {
    const uint8_t b = d[i];
    out[index++] = b;

    if (info->rle[b])
    {
      /* :497-511 (range 255) and :521-535 (range min(n - i - 1, 255) in the last 256 bytes) */
      const uint32_t left = n - i - 1, target = (n >= 256) ? n - 256 : 0;
      const uint32_t range = (i < target) ? maxLen : (left < maxLen ? left : maxLen);   /* (Short form: 32, rle8_low_entropy_short_cpu.c:152, :179) */
      uint32_t count = 0, j = 1;

      for (; j < range; j++)
        if (d[i + j] == b) count++; else break;

      i += j - 1;
      out[index++] = info->symbolsByProb[count & 0xFF];
    }
  }

  return index;
}

static uint32_t le_compress_section(const uint8_t *d, uint32_t n, const le_info_t *info, uint8_t *out) { return le_compress_section_ex(d, n, info, out, 255); }

uint32_t hso_rle8m_compress(uint32_t subSections, const uint8_t *pIn, uint32_t inSize, uint8_t *pOut, uint32_t outSize)
{
  if (pIn == NULL || inSize == 0 || pOut == NULL || subSections == 0 || outSize < hso_rle8m_compress_bounds(subSections, inSize))
    return 0;

  le_info_t info;
  le_get_info(pIn, inSize, &info);

  uint32_t index = 4;
  patch32(pOut, index, inSize); index += 4;
  patch32(pOut, index, subSections); index += 4;
  const uint32_t table = index;
  index += 4 * (subSections - 1);
  index += le_write_info(&info, pOut + index);

  const uint32_t ss = inSize / subSections;

  for (uint32_t i = 0; i + 1 < subSections; i++)
  {
    if (ss == 0 || outSize - index < ss) return 0;  /* compress_with_info fails on an empty section and when the room left is below the section size (:476) */
    index += le_compress_section(pIn + ss * i, ss, &info, pOut + index);
    patch32(pOut, table + 4 * i, index);
  }

  if (outSize - index < inSize - ss * (subSections - 1)) return 0;
  index += le_compress_section(pIn + ss * (subSections - 1), inSize - ss * (subSections - 1), &info, pOut + index);
  patch32(pOut, 0, index);
  return index;
}

uint32_t hso_rle8m_decompress(const uint8_t *pIn, uint32_t inSize, uint8_t *pOut, uint32_t outSize)
{
  if (pIn == NULL || pOut == NULL || inSize < 12 || outSize == 0)
    return 0;

  const uint32_t expIn = get32(pIn), expOut = get32(pIn + 4), subSections = get32(pIn + 8);

  if (expOut > outSize || expIn > inSize || subSections == 0)
    return 0;

  uint64_t index = 12 + 4ull * (subSections - 1);
  if (index + 33 > expIn) return 0;

  uint8_t rle[256], symbolToCount[256], listed[256];
  for (int i = 0; i < 256; i++) rle[i] = (pIn[index + (i >> 3)] >> (i & 7)) & 1;
  index += 32;
  uint32_t sc = pIn[index++];
  if (!sc) sc = 255;
  if (index + sc > expIn) return 0;
  memset(listed, 0, 256);
  for (uint32_t i = 0; i < sc; i++) { symbolToCount[pIn[index + i]] = (uint8_t)i; listed[pIn[index + i]] = 1; }
  index += sc;
  uint32_t next = sc;
  for (int i = 0; i < 256; i++) if (!listed[i]) symbolToCount[i] = (uint8_t)next++;

  const uint32_t ss = expOut / subSections;
  uint64_t o = 0;

  for (uint32_t k = 0; k < subSections; k++)
  {
    const uint64_t end = (k + 1 < subSections) ? get32(pIn + 12 + 4 * k) : expIn;
    const uint64_t want = (k + 1 < subSections) ? ss : expOut - (uint64_t)ss * (subSections - 1);
    const uint64_t oEnd = o + want;
    if (end > expIn || end < index) return 0;

    while (index < end)
    {
      const uint8_t b = pIn[index++];
      if (o >= oEnd) return 0;
      pOut[o++] = b;

      if (rle[b])
      {
        if (index >= end) return 0;
        uint32_t count = symbolToCount[pIn[index++]];
        if (o + count > oEnd) return 0;
        memset(pOut + o, b, count);
        o += count;
      }
    }

    if (o != oEnd) return 0;
  }

  return expOut;
}

/* The UNSECTIONED forms of the same codec: [u32 compressedSize][u32 uncompressedSize][info][one stream].
 *   rle8_low_entropy_compress / _compress_only_max_frequency / _decompress            src/rle8_low_entropy_cpu.c:6-124
 *   rle8_low_entropy_short_compress / _only_max_frequency / _short_decompress         src/rle8_low_entropy_short_cpu.c:16-124
 * variant bit 0: the Short form (runs cut every 32 bytes instead of 255), bit 1: only_max_frequency.  Like the reference, the encoder
 * does not check the room its stream needs beyond `bounds` (it can take up to twice the input): callers give it 2 * inSize more. */
uint32_t hso_low_entropy_compress_bounds(uint32_t inSize) { return inSize + (256 / 8) + 1 + 256 + 4 * 2; }

uint32_t hso_low_entropy_compress(int variant, const uint8_t *pIn, uint32_t inSize, uint8_t *pOut, uint32_t outSize)
{
  if (pIn == NULL || inSize == 0 || pOut == NULL || outSize < hso_low_entropy_compress_bounds(inSize))
    return 0;

  const uint32_t maxLen = (variant & 1) ? 32 : 255;
  le_info_t info;
  le_get_info_ex(pIn, inSize, &info, maxLen, (variant & 2) != 0);

  uint32_t index = 4;
  patch32(pOut, index, inSize); index += 4;
  index += le_write_info(&info, pOut + index);
  if (outSize - index < inSize) return 0;                      /* compress_with_info: outSize < inSize (:476) -- never true above the bound */
  index += le_compress_section_ex(pIn, inSize, &info, pOut + index, maxLen);
  patch32(pOut, 0, index);
  return index;
}

/* one decoder for both forms: the count comes out of the code table (src/rle8_low_entropy_cpu.c:96-124, :930-1021; short: :103-124, :454-534) */
uint32_t hso_low_entropy_decompress(const uint8_t *pIn, uint32_t inSize, uint8_t *pOut, uint32_t outSize)
{
  if (pIn == NULL || pOut == NULL || inSize < 8 || outSize == 0)
    return 0;

  const uint32_t expIn = get32(pIn), expOut = get32(pIn + 4);

  if (expOut > outSize || expIn > inSize)
    return 0;

  uint64_t index = 8;
  if (index + 33 > expIn) return 0;

  uint8_t rle[256], symbolToCount[256], listed[256];
  for (int i = 0; i < 256; i++) rle[i] = (pIn[index + (i >> 3)] >> (i & 7)) & 1;
  index += 32;
  uint32_t sc = pIn[index++];
  if (!sc) sc = 255;
  if (index + sc > expIn) return 0;
  memset(listed, 0, 256);
  for (uint32_t i = 0; i < sc; i++) { symbolToCount[pIn[index + i]] = (uint8_t)i; listed[pIn[index + i]] = 1; }
  index += sc;
  uint32_t next = sc;
  for (int i = 0; i < 256; i++) if (!listed[i]) symbolToCount[i] = (uint8_t)next++;

  uint64_t o = 0;
  while (index < expIn)
  {
    const uint8_t b = pIn[index++];
    if (o >= expOut) return 0;
    pOut[o++] = b;
    if (rle[b])
    {
      if (index >= expIn) return 0;
      const uint32_t count = symbolToCount[pIn[index++]];
      if (o + count > expOut) return 0;
      memset(pOut + o, b, count);
      o += count;
    }
  }
  return (o == expOut) ? expOut : 0;
}

/* ------------------------------------------------------------------------------------------ */
/* public entry points                                                                        */

static int hso_short_k(int family) { return (family == HSO_SHORT0 || family == HSO_SINGLE_SHORT) ? 0 : ((family == HSO_SHORT1 || family == HSO_GREEDY1) ? 1 : ((family == HSO_SHORT3 || family == HSO_GREEDY3) ? 3 : 7)); }
static int valid_S(int S) { return S == 1 || S == 2 || S == 3 || S == 4 || S == 6 || S == 8 || S == 16; }

uint32_t hso_compress(int family, int S, int aligned, const uint8_t *pIn, uint32_t inSize, uint8_t *pOut, uint32_t outSize)
{
  /* argument checks: rle8_extreme_cpu.h:88-89, rleX_extreme_cpu.h:49-50, rleX_Xsl.h:271-272 */
  if (pIn == NULL || inSize == 0 || pOut == NULL || outSize < hso_compress_bounds(inSize) || !valid_S(S))
    return 0;
  if (inSize > (1u << 30)) /* q8: treat > 1 GiB single-stream input as unsupported */
    return 0;

  switch (family)
  {
  case HSO_PLAIN:
  case HSO_PACKED:
    if (S == 1) return enc8_multi(pIn, inSize, family == HSO_PACKED, pOut);
    if (S == 16) return enc128(pIn, inSize, aligned, family == HSO_PACKED, pOut);
    return enc_generic(pIn, inSize, S, aligned, family == HSO_PACKED, pOut);

  case HSO_LUT3:
  case HSO_LUT7:
    if (S == 16) return 0;
    return enc_lut(pIn, inSize, S, aligned, family == HSO_LUT3 ? 3 : 7, pOut);

  case HSO_SINGLE:
  case HSO_PACKED_SINGLE:
    if (S != 1) return 0;
    return enc8_single(pIn, inSize, family == HSO_PACKED_SINGLE, pOut);

  case HSO_SHORT0:
  case HSO_SHORT1:
  case HSO_SHORT3:
  case HSO_SHORT7:
    if (S == 16) return 0;
    return enc_short(pIn, inSize, S, aligned, hso_short_k(family), pOut);

  case HSO_SINGLE_SHORT:
    if (S != 1) return 0;
    return enc_single_short(pIn, inSize, pOut);

  case HSO_GREEDY1:
  case HSO_GREEDY3:
  case HSO_GREEDY7:
    if (S == 1 || S == 16 || aligned) return 0;
    return enc_short_greedy(pIn, inSize, S, hso_short_k(family), pOut);
  }

  return 0;
}

uint32_t hso_decompress(int family, int S, int aligned, const uint8_t *pIn, uint32_t inSize, uint8_t *pOut, uint32_t outSize)
{
  /* argument + header checks: rle8_extreme_cpu.h:704-712, rleX_extreme_cpu.h:84-91, rleX_Xsl.h:1850-1858 */
  if (pIn == NULL || pOut == NULL || inSize == 0 || outSize == 0 || !valid_S(S))
    return 0;

  const int lut = family == HSO_LUT3 || family == HSO_LUT7;
  const int shortFam = family >= HSO_SHORT0 && family <= HSO_SINGLE_SHORT;   /* Greedy streams are Short streams */
  const size_t headerSize = (S == 1 && !lut && !shortFam) ? 9 : 8;

  if (inSize < headerSize)
    return 0;

  const uint32_t U = get32(pIn), C = get32(pIn + 4);

  if (U > outSize || C > inSize || C < headerSize)
    return 0;

  src_t s = { pIn, headerSize, C, 0 };
  dst_t o = { pOut, 0, U, 0 };
  int ok;

  if (lut)
  {
    if (S == 16) return 0;
    ok = dec_lut(&s, &o, S, aligned, family == HSO_LUT3 ? 3 : 7);
  }
  else if (shortFam)
  {
    if (S == 16) return 0;
    ok = dec_short(&s, &o, S, aligned, hso_short_k(family), family == HSO_SINGLE_SHORT);
  }
  else
  {
    const int packed = family == HSO_PACKED || family == HSO_PACKED_SINGLE;
    uint8_t sym[16];
    memset(sym, 0, sizeof(sym));

    if (S == 1)
    {
      const uint8_t mode = pIn[8];

      if (mode == 0)
        ok = dec_packets(&s, &o, 1, 0, packed, packed, 0, packed ? 3 : 6, sym);
      else if (mode == 1)
      {
        sym[0] = (uint8_t)rd8(&s);
        ok = dec_packets(&s, &o, 1, 0, 0, 0, 1, packed ? 2 : 4, sym);
      }
      else
        return 0;
    }
    else
    {
      ok = dec_packets(&s, &o, S, aligned, packed, packed && !aligned, 0, packed ? 3 : (uint32_t)S + 4, sym);
    }
  }

  if (!ok || o.at != U)
    return 0;

  return U;
}

/* name table: src/rle.h:100-394 */
int hso_resolve(const char *name, int *family, int *S, int *aligned, int *isDecompress)
{
  static const struct { const char *w; int S; } widths[] = { { "rle16_", 2 }, { "rle24_", 3 }, { "rle32_", 4 }, { "rle48_", 6 }, { "rle64_", 8 }, { "rle128_", 16 } };
  static const struct { const char *n; int fam, dec; } eight[] = {
    { "rle8_multi_compress", HSO_PLAIN, 0 }, { "rle8_single_compress", HSO_SINGLE, 0 }, { "rle8_decompress", HSO_PLAIN, 1 },
    { "rle8_packed_multi_compress", HSO_PACKED, 0 }, { "rle8_packed_single_compress", HSO_PACKED_SINGLE, 0 }, { "rle8_packed_decompress", HSO_PACKED, 1 },
    { "rle8_3symlut_compress", HSO_LUT3, 0 }, { "rle8_3symlut_decompress", HSO_LUT3, 1 },
    { "rle8_7symlut_compress", HSO_LUT7, 0 }, { "rle8_7symlut_decompress", HSO_LUT7, 1 },
    { "rle8_single_short_compress", HSO_SINGLE_SHORT, 0 }, { "rle8_single_short_decompress", HSO_SINGLE_SHORT, 1 },
    { "rle8_multi_short_compress", HSO_SHORT0, 0 }, { "rle8_multi_short_decompress", HSO_SHORT0, 1 },
    { "rle8_1symlut_short_compress", HSO_SHORT1, 0 }, { "rle8_1symlut_short_decompress", HSO_SHORT1, 1 },
    { "rle8_3symlut_short_compress", HSO_SHORT3, 0 }, { "rle8_3symlut_short_decompress", HSO_SHORT3, 1 },
    { "rle8_7symlut_short_compress", HSO_SHORT7, 0 }, { "rle8_7symlut_short_decompress", HSO_SHORT7, 1 },
  };
  static const struct { const char *mid; int fam, aligned; } mids[] = {
    { "sym_short_", HSO_SHORT0, 1 }, { "byte_short_", HSO_SHORT0, 0 },
    { "1symlut_sym_short_", HSO_SHORT1, 1 }, { "1symlut_byte_short_", HSO_SHORT1, 0 },
    { "3symlut_sym_short_", HSO_SHORT3, 1 }, { "3symlut_byte_short_", HSO_SHORT3, 0 },
    { "7symlut_sym_short_", HSO_SHORT7, 1 }, { "7symlut_byte_short_", HSO_SHORT7, 0 },
    { "sym_packed_", HSO_PACKED, 1 }, { "byte_packed_", HSO_PACKED, 0 }, { "sym_", HSO_PLAIN, 1 }, { "byte_", HSO_PLAIN, 0 },
    { "3symlut_sym_", HSO_LUT3, 1 }, { "3symlut_byte_", HSO_LUT3, 0 }, { "7symlut_sym_", HSO_LUT7, 1 }, { "7symlut_byte_", HSO_LUT7, 0 },
  };

  for (size_t k = 0; k < sizeof(eight) / sizeof(eight[0]); k++)
    if (strcmp(name, eight[k].n) == 0)
    {
      *family = eight[k].fam; *S = 1; *aligned = 0; *isDecompress = eight[k].dec;
      return 1;
    }

  for (size_t w = 0; w < sizeof(widths) / sizeof(widths[0]); w++)
  {
    const size_t wl = strlen(widths[w].w);
    if (strncmp(name, widths[w].w, wl) != 0)
      continue;

    for (size_t m = 0; m < sizeof(mids) / sizeof(mids[0]); m++)
    {
      const size_t ml = strlen(mids[m].mid);
      if (strncmp(name + wl, mids[m].mid, ml) != 0)
        continue;

      const char *tail = name + wl + ml;
      int dec, fam = mids[m].fam;
      if (strcmp(tail, "compress") == 0) dec = 0;
      else if (strcmp(tail, "decompress") == 0) dec = 1;
      else if (strcmp(tail, "compress_greedy") == 0 && !mids[m].aligned && fam >= HSO_SHORT1 && fam <= HSO_SHORT7)
      {
        dec = 0;
        fam = fam == HSO_SHORT1 ? HSO_GREEDY1 : (fam == HSO_SHORT3 ? HSO_GREEDY3 : HSO_GREEDY7);   /* src/rle.h:398-416 */
      }
      else continue;

      if (widths[w].S == 16 && (mids[m].fam == HSO_LUT3 || mids[m].fam == HSO_LUT7 || mids[m].fam >= HSO_SHORT0))
        return 0;

      *family = fam; *S = widths[w].S; *aligned = mids[m].aligned; *isDecompress = dec;
      return 1;
    }
  }

  return 0;
}

uint32_t hso_call(const char *name, const uint8_t *pIn, uint32_t inSize, uint8_t *pOut, uint32_t outSize)
{
  int fam, S, al, dec;
  if (!hso_resolve(name, &fam, &S, &al, &dec))
    return 0xFFFFFFFFu;
  return dec ? hso_decompress(fam, S, al, pIn, inSize, pOut, outSize) : hso_compress(fam, S, al, pIn, inSize, pOut, outSize);
}

uint32_t hso_compress_blocks(int family, int S, int aligned, const uint8_t *pIn, uint64_t inSize, uint32_t blockSize,
                             uint8_t *pOut, uint32_t stride, uint32_t *pSizes)
{
  if (blockSize == 0 || stride < hso_compress_bounds(blockSize))
    return 0;

  const uint64_t nBlocks = (inSize + blockSize - 1) / blockSize;

  for (uint64_t b = 0; b < nBlocks; b++)
  {
    const uint64_t off = b * blockSize;
    const uint32_t len = (uint32_t)((inSize - off) < blockSize ? (inSize - off) : blockSize);
    const uint32_t c = hso_compress(family, S, aligned, pIn + off, len, pOut + b * stride, stride);
    if (c == 0)
      return 0;
    pSizes[b] = c;
  }

  return (uint32_t)nBlocks;
}

/* Decode `nBlocks` block streams (payload + offsets[i] .. offsets[i+1]) into pOut + i * blockSize.  Returns the number of
 * bytes produced, 0 on any failure.  Used by tests and by bench.py's cpu_baseline leg ("port"). */
uint64_t hso_decompress_blocks(int family, int S, int aligned, const uint8_t *payload, const uint64_t *offsets, uint64_t nBlocks,
                               uint32_t blockSize, uint8_t *pOut, uint64_t outSize)
{
  uint64_t produced = 0;

  for (uint64_t b = 0; b < nBlocks; b++)
  {
    const uint64_t at = b * blockSize;
    if (at >= outSize)
      return 0;
    const uint32_t room = (uint32_t)((outSize - at) < blockSize ? (outSize - at) : blockSize);
    const uint32_t got = hso_decompress(family, S, aligned, payload + offsets[b], (uint32_t)(offsets[b + 1] - offsets[b]), pOut + at, room);
    if (got == 0)
      return 0;
    produced += got;
  }

  return produced;
}

/* ---- big-config manifests: hash of every block stream of the ORACLE's encoder (oracle/hsrle_hash.h), to be compared with the roll-ups
 *      the compiled reference minted (tests/golden/big/, tests/test_oracle_golden.py) ---- */
#include "hsrle_hash.h"
#include <stdlib.h>

uint64_t hso_hash64(const uint8_t *p, uint64_t len) { return hsrle_hash64(p, len); }

uint64_t hso_hash_blocks(int family, int S, int aligned, const uint8_t *pIn, uint64_t inSize, uint32_t blockSize, uint64_t *pHashes, uint32_t *pSizes)
{
  const uint32_t stride = hso_compress_bounds(blockSize) + 64;
  uint8_t *tmp = (uint8_t *)malloc(stride);
  if (!tmp || blockSize == 0) { free(tmp); return 0; }
  const uint64_t nBlocks = (inSize + blockSize - 1) / blockSize;
  for (uint64_t b = 0; b < nBlocks; b++)
  {
    const uint64_t off = b * blockSize;
    const uint32_t len = (uint32_t)((inSize - off) < blockSize ? (inSize - off) : blockSize);
    const uint32_t c = hso_compress(family, S, aligned, pIn + off, len, tmp, stride);
    if (c == 0) { free(tmp); return 0; }
    pHashes[b] = hsrle_hash64(tmp, c);
    if (pSizes) pSizes[b] = c;
  }
  free(tmp);
  return nBlocks;
}

void hso_rollups(const uint64_t *hashes, uint64_t n, uint64_t group, uint64_t *out)
{
  for (uint64_t g = 0; g * group < n; g++) out[g] = hsrle_rollup(hashes + g * group, (n - g * group) < group ? (n - g * group) : group);
}
