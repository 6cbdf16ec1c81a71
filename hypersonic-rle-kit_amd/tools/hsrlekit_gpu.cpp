// hsrlekit_gpu -- benchmark table of the rleX_extreme codecs on the GPU, over the C ABI of libhsrle_hip.so (include/hsrle.h).
//
// The counterpart of the reference's `hsrlekit <file>` benchmark (protocol: src/main.c:803-1076 -- per codec: compress `runs`
// times, decompress `runs` times, validate, one table row with ratio, mean and best speed), with the device-resident block
// container API instead of the host functions.  Own code; the row layout follows the reference's table so that the two can be
// read side by side:
//     <mode>                        | <ratio> % | <encode mean> (<best>) | <decode mean> (<best>)
// Speeds are GiB/s of uncompressed bytes, timed with HIP events on the launch stream (input and output stay in HBM).
//
//   hsrlekit_gpu <file> | --synth runs|video <MiB>   [--runs N] [--block B] [--codec <rle.h name>] [--host]
//       --host   additionally time the host-pointer drop-in functions (rle.h names: one monolithic stream, PCIe included)
//
// Build: make -C hypersonic-rle-kit_amd tools      Exit code: 0 = every codec round-tripped, 1 = failure.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "hsrle.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

// display name in the reference's style (src/codec_funcs.h:135-258): "8 Bit Packed", "16 Bit 3LUT (Symbol)", ...
static std::string display_name(int codec)
{
  const std::string n = hsrle_codec_name(codec);           // e.g. rle16_3symlut_byte, rle8_packed_single, rle64_sym_packed
  const int bits = atoi(n.c_str() + 3);
  std::string s = std::to_string(bits) + " Bit";
  if (n.find("3symlut") != std::string::npos) s += " 3LUT";
  if (n.find("7symlut") != std::string::npos) s += " 7LUT";
  if (n.find("single") != std::string::npos) s += " Single";
  if (n.find("packed") != std::string::npos) s += " Packed";
  if (bits != 8) s += (n.find("byte") != std::string::npos) ? " (Byte)" : " (Symbol)";
  s.resize(30, ' ');
  return s;
}

typedef uint32_t (*dropin_fn)(const uint8_t *, uint32_t, uint8_t *, uint32_t);

int main(int argc, char **argv)
{
  const char *file = nullptr, *synth = nullptr, *only = nullptr;
  uint64_t synthMiB = 0;
  int runs = 10;
  uint32_t block = 4096;
  bool host = false;

  for (int i = 1; i < argc; i++)
  {
    if (!strcmp(argv[i], "--synth") && i + 2 < argc) { synth = argv[i + 1]; synthMiB = strtoull(argv[i + 2], nullptr, 10); i += 2; }
    else if (!strcmp(argv[i], "--runs") && i + 1 < argc) runs = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--block") && i + 1 < argc) block = (uint32_t)atoi(argv[++i]);
    else if (!strcmp(argv[i], "--codec") && i + 1 < argc) only = argv[++i];
    else if (!strcmp(argv[i], "--host")) host = true;
    else if (argv[i][0] != '-') file = argv[i];
    else { fprintf(stderr, "unknown argument '%s'\n", argv[i]); return 2; }
  }
  if ((!file && !synth) || runs < 1)
  {
    fprintf(stderr, "usage: hsrlekit_gpu <file> | --synth runs|video <MiB>  [--runs N] [--block B] [--codec NAME] [--host]\n");
    return 2;
  }
  if (hsrle_device_count() < 1) { fprintf(stderr, "no usable HIP device (there is no CPU fallback)\n"); return 2; }

  // ---- input ----
  std::vector<uint8_t> hostIn;
  uint64_t size = 0;
  if (file)
  {
    FILE *f = fopen(file, "rb");
    if (!f) { fprintf(stderr, "cannot open '%s'\n", file); return 2; }
    fseek(f, 0, SEEK_END); size = (uint64_t)ftell(f); fseek(f, 0, SEEK_SET);
    hostIn.resize(size);
    if (size == 0 || fread(hostIn.data(), 1, size, f) != size) { fprintf(stderr, "cannot read '%s'\n", file); return 2; }
    fclose(f);
  }
  else
    size = synthMiB << 20;

  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));
  const uint64_t bound = hsrle_container_bound(size, block), wsz = hsrle_compress_workspace_size(size, block);
  uint8_t *dIn, *dCont, *dOut, *dWs;
  uint32_t *dStatus;
  HIP_OK(hipMalloc((void **)&dIn, size)); HIP_OK(hipMalloc((void **)&dCont, bound)); HIP_OK(hipMalloc((void **)&dOut, size));
  HIP_OK(hipMalloc((void **)&dWs, wsz)); HIP_OK(hipMalloc((void **)&dStatus, 64));
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));

  printf("\nBenchmarking %s (%llu Bytes), %u byte blocks, %d runs, device resident\n\n", file ? file : (std::string("synthetic ") + synth).c_str(),
         (unsigned long long)size, block, runs);
  printf("Mode                          | Compression Rate | Compression Speed (best)      | Decompression Speed (best)\n");
  printf("------------------------------------------------------------------------------------------------------------------\n");

  int failures = 0;
  int lastS = -1;
  std::vector<uint8_t> back;

  for (int codec = 0; codec < HSRLE_CODEC_COUNT; codec++)
  {
    const std::string name = hsrle_codec_name(codec);
    if (only && name != only) continue;
    const int bits = atoi(name.c_str() + 3), S = bits / 8;

    if (file) { if (lastS < 0) HIP_OK(hipMemcpy(dIn, hostIn.data(), size, hipMemcpyHostToDevice)); lastS = 0; }
    else if (S != lastS)
    {
      // synthetic input of this symbol width (SURVEY.md §8d generators, seed 5 as in BASELINE config 5)
      if (hsrle_synth_dev_async(!strcmp(synth, "video") ? HSRLE_SYNTH_VIDEO : HSRLE_SYNTH_RUNS, S, 5, dIn, size, st) != 0) { fprintf(stderr, "synth failed\n"); return 2; }
      HIP_OK(hipStreamSynchronize(st));
      lastS = S;
    }

    // ---- compress: one warm-up (discarded, like the reference's dry run), then `runs` timed ----
    double encSum = 0, encBest = 1e30, decSum = 0, decBest = 1e30;
    bool ok = true;
    hsrle_container_info_t info;
    for (int r = -1; r < runs && ok; r++)
    {
      HIP_OK(hipEventRecord(e0, st));
      ok = hsrle_compress_dev_async(codec, dIn, size, dCont, bound, block, dWs, wsz, st) == 0;
      HIP_OK(hipEventRecord(e1, st));
      HIP_OK(hipEventSynchronize(e1));
      float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 0) { encSum += ms; encBest = std::min(encBest, (double)ms); }
    }
    ok = ok && hsrle_container_info_dev(dCont, bound, &info, st) == 0 && info.uncompressedSize == size;

    // ---- decompress ----
    HIP_OK(hipMemsetAsync(dOut, 0xEE, size, st));
    for (int r = -1; r < runs && ok; r++)
    {
      HIP_OK(hipMemsetAsync(dStatus, 0, 4, st));
      HIP_OK(hipEventRecord(e0, st));
      ok = hsrle_decompress_dev_async(dCont, &info, dOut, size, dStatus, st) == 0;
      HIP_OK(hipEventRecord(e1, st));
      HIP_OK(hipEventSynchronize(e1));
      float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 0) { decSum += ms; decBest = std::min(decBest, (double)ms); }
    }

    // ---- validate: status word and a byte compare of the whole buffer (on the host, like the reference's memcmp) ----
    if (ok)
    {
      uint32_t status = 1;
      HIP_OK(hipMemcpy(&status, dStatus, 4, hipMemcpyDeviceToHost));
      ok = status == 0;
      if (ok)
      {
        back.resize(size);
        HIP_OK(hipMemcpy(back.data(), dOut, size, hipMemcpyDeviceToHost));
        if (hostIn.size() != size || lastS != 0) { hostIn.resize(size); HIP_OK(hipMemcpy(hostIn.data(), dIn, size, hipMemcpyDeviceToHost)); }
        ok = memcmp(back.data(), hostIn.data(), size) == 0;
      }
    }

    const double gib = (double)size / (double)(1ull << 30);
    if (!ok)
    {
      printf("%s| <FAILED>\n", display_name(codec).c_str());
      failures++;
      continue;
    }
    printf("%s| %14.2f %% | %8.1f GiB/s (%8.1f GiB/s) | %8.1f GiB/s (%8.1f GiB/s)", display_name(codec).c_str(), 100.0 * (double)info.totalSize / (double)size,
           gib / (encSum / runs / 1e3), gib / (encBest / 1e3), gib / (decSum / runs / 1e3), gib / (decBest / 1e3));

    if (host && size <= (1ull << 30))
    {
      // the drop-in functions of rle.h: host pointers, one monolithic reference stream, resolved by name like a linker would
      const uint32_t cap = rle_compress_bounds((uint32_t)size);
      std::vector<uint8_t> comp(cap), out2(size + rle_decompress_additional_size());
      const int cid = codec;
      hipEvent_t h0, h1; HIP_OK(hipEventCreate(&h0)); HIP_OK(hipEventCreate(&h1));
      HIP_OK(hipEventRecord(h0, st));
      const uint32_t csz = hsrle_compress_mono(cid, hostIn.data(), (uint32_t)size, comp.data(), cap);
      HIP_OK(hipEventRecord(h1, st)); HIP_OK(hipEventSynchronize(h1));
      float cms; HIP_OK(hipEventElapsedTime(&cms, h0, h1));
      HIP_OK(hipEventRecord(h0, st));
      const uint32_t usz = csz ? hsrle_decompress_mono(cid, comp.data(), csz, out2.data(), (uint32_t)size) : 0;
      HIP_OK(hipEventRecord(h1, st)); HIP_OK(hipEventSynchronize(h1));
      float dms; HIP_OK(hipEventElapsedTime(&dms, h0, h1));
      const bool hok = usz == size && memcmp(out2.data(), hostIn.data(), size) == 0;
      printf(" | drop-in: %6.2f %%, %7.1f / %7.1f MiB/s%s", 100.0 * csz / (double)size, (size / 1048576.0) / (cms / 1e3), (size / 1048576.0) / (dms / 1e3), hok ? "" : " <FAILED>");
      if (!hok) failures++;
      HIP_OK(hipEventDestroy(h0)); HIP_OK(hipEventDestroy(h1));
    }
    printf("\n");
    fflush(stdout);
  }

  printf("\n%s\n", failures ? "FAILED" : "all codecs round-tripped");
  return failures ? 1 : 0;
}
