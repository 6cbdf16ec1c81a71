/*
 * hsrle_synth.c -- CPU statement of the deterministic synthetic workloads (TEST / BENCH INFRASTRUCTURE ONLY).
 *
 * The same generators exist on the device (hsrle_synth_dev_async in the product library) and in pure python
 * (tests/hsrle_testlib.py:synth_chunk_py); tests check that all three produce identical bytes.  Definition
 * (made chunk-parallel): the buffer is cut into 64 KiB chunks; chunk c is generated from a splitmix64
 * stream seeded with seed * 0x9E3779B97F4A7C15 + c * 0xD1B54A32D192ED03 + kind and truncated to the chunk size.
 *   kind 0 (run-distributed, symbol width S): repeat { L = 1 + r % 63 PRNG bytes; R = 2 + r % 62 copies of one S-byte PRNG symbol }
 *   kind 1 (video-frame-shaped):              repeat { Z zero bytes, Z = 40 + r % 120 for one r in four else 10 + r % 16;
 *                                                      1 + r % 9 bytes out of {01,02,03,FF,FE,04} }   (calibrated: rle8_packed ~16 %, rle64_3symlut_byte ~22 %)
 */
#include <stdint.h>
#include <stddef.h>

#define HSO_SYNTH_CHUNK 65536u

static uint64_t splitmix64(uint64_t *state)
{
  *state += 0x9E3779B97F4A7C15ull;
  uint64_t z = *state;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

static void synth_chunk(int kind, int S, uint64_t seed, uint64_t chunk, uint8_t *o, uint32_t len)
{
  uint64_t st = seed * 0x9E3779B97F4A7C15ull + chunk * 0xD1B54A32D192ED03ull + (uint64_t)kind;
  uint32_t at = 0;

  if (kind == 0)
  {
    while (at < len)
    {
      uint64_t r = splitmix64(&st);
      const uint32_t L = 1u + (uint32_t)(r % 63u);
      for (uint32_t k = 0; k < L; k += 8)
      {
        const uint64_t v = splitmix64(&st);
        for (uint32_t j = 0; j < 8 && k + j < L; j++)
          if (at + k + j < len) o[at + k + j] = (uint8_t)(v >> (8 * j));
      }
      at += L;

      r = splitmix64(&st);
      const uint32_t R = 2u + (uint32_t)(r % 62u);
      uint8_t sym[16];
      for (int k = 0; k < S; k += 8)
      {
        const uint64_t v = splitmix64(&st);
        for (int j = 0; j < 8 && k + j < S; j++) sym[k + j] = (uint8_t)(v >> (8 * j));
      }
      for (uint32_t k = 0; k < R * (uint32_t)S && at + k < len; k++)
        o[at + k] = sym[k % (uint32_t)S];
      at += R * (uint32_t)S;
    }
  }
  else
  {
    static const uint8_t vals[6] = { 0x01, 0x02, 0x03, 0xFF, 0xFE, 0x04 };
    while (at < len)
    {
      uint64_t r = splitmix64(&st);
      const uint32_t Z = (((r >> 32) & 3u) == 0u) ? 40u + (uint32_t)(r % 120u) : 10u + (uint32_t)(r % 16u);
      for (uint32_t k = 0; k < Z && at + k < len; k++) o[at + k] = 0;
      at += Z;
      r = splitmix64(&st);
      const uint32_t Bn = 1u + (uint32_t)(r % 9u);
      for (uint32_t k = 0; k < Bn && at + k < len; k++) o[at + k] = vals[(r >> (8 + 4 * k)) % 6u];
      at += Bn;
    }
  }
}

/* Fill out[0, size) with bytes [offset, offset + size) of the synthetic buffer; offset must be a multiple of 64 KiB. */
int hso_synth(int kind, int symbolBytes, uint64_t seed, uint64_t offset, uint8_t *out, uint64_t size)
{
  if (!out || (offset % HSO_SYNTH_CHUNK) != 0 || symbolBytes < 1 || symbolBytes > 16 || (kind != 0 && kind != 1))
    return 0;
  for (uint64_t done = 0; done < size; done += HSO_SYNTH_CHUNK)
  {
    const uint32_t len = (uint32_t)((size - done) < HSO_SYNTH_CHUNK ? (size - done) : HSO_SYNTH_CHUNK);
    synth_chunk(kind, symbolBytes, seed, (offset + done) / HSO_SYNTH_CHUNK, out + done, len);
  }
  return 1;
}
