/*
 * hsrle_hash.h -- the 64 bit block-stream hash of the big-config manifests (TEST INFRASTRUCTURE; the product library has the same
 * function as a device kernel, csrc/hsrle_capi.hip k_hash_blocks, declared in include/hsrle.h).
 *
 *   h = 0x9E3779B97F4A7C15 ^ (len * 0xD6E8FEB86659FD93)
 *   for every 8-byte little-endian word w of the bytes (the last one zero padded):  h = rotl64(h ^ w, 27) * 0x9E3779B97F4A7C15 + 0x165667B19E3779F9
 *   h ^= h >> 31
 *
 * Roll-up of a group of block hashes b0, b1, ... (in order):  r = 0;  r = (rotl64(r, 7) ^ b) * 0x9E3779B97F4A7C15
 */
#ifndef HSRLE_HASH_H
#define HSRLE_HASH_H
#include <stdint.h>
#include <string.h>

static inline uint64_t hsrle_rotl64(uint64_t v, int s) { return (v << s) | (v >> (64 - s)); }

static inline uint64_t hsrle_hash64(const uint8_t *p, uint64_t len)
{
  uint64_t h = 0x9E3779B97F4A7C15ull ^ (len * 0xD6E8FEB86659FD93ull);
  uint64_t k = 0;
  for (; k + 8 <= len; k += 8)
  {
    uint64_t w;
    memcpy(&w, p + k, 8);
    h = hsrle_rotl64(h ^ w, 27) * 0x9E3779B97F4A7C15ull + 0x165667B19E3779F9ull;
  }
  if (k < len)
  {
    uint64_t w = 0;
    memcpy(&w, p + k, (size_t)(len - k));
    h = hsrle_rotl64(h ^ w, 27) * 0x9E3779B97F4A7C15ull + 0x165667B19E3779F9ull;
  }
  return h ^ (h >> 31);
}

static inline uint64_t hsrle_rollup(const uint64_t *hashes, uint64_t n)
{
  uint64_t r = 0;
  for (uint64_t i = 0; i < n; i++) r = (hsrle_rotl64(r, 7) ^ hashes[i]) * 0x9E3779B97F4A7C15ull;
  return r;
}
#endif
