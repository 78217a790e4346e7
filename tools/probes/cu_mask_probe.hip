// Probe of hipExtStreamCreateWithCUMask on MI355X (8 XCDs x 32 CUs): which (XCC, SE, CU) the workgroups of a census kernel land on
// for a few masks — does bit i mean "CU i % 32 of XCD i / 32", "CU i / 8 of XCD i % 8", or something else; and can a stream be
// confined to a handful of CUs per XCD (the tracker's window chain beside the ViT encoder).
//   hipcc --offload-arch=gfx950 -O2 -o cu_mask_probe cu_mask_probe.hip && ./cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <map>
#include <set>
#include <vector>

__global__ void census(unsigned* out, long spin) {
  const long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) {}
  if (threadIdx.x == 0) {
    // s_getreg_b32: simm16 = (size - 1) << 11 | offset << 6 | id;  HW_REG_HW_ID = 4, HW_REG_XCC_ID = 20
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);
    out[blockIdx.x * 2] = hw, out[blockIdx.x * 2 + 1] = xcc;
  }
}

static void run(const char* name, const std::vector<uint32_t>& mask, unsigned* d, int nblk) {
  hipStream_t s;
  hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
  if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask failed: %s\n", name, hipGetErrorString(e)); return; }
  hipMemsetAsync(d, 0xff, nblk * 8, s);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  hipEventRecord(e0, s);
  hipLaunchKernelGGL(census, dim3(nblk), dim3(256), 0, s, d, 2000L);   // 2000 ticks of the 100 MHz wall clock = 20 us per block
  hipEventRecord(e1, s);
  hipStreamSynchronize(s);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned> h(nblk * 2);
  hipMemcpy(h.data(), d, nblk * 8, hipMemcpyDeviceToHost);
  std::map<unsigned, std::set<unsigned>> per_xcc;     // xcc -> set of (se, sh, cu)
  for (int b = 0; b < nblk; ++b) {
    const unsigned hw = h[b * 2], xcc = h[b * 2 + 1] & 0xf;
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    per_xcc[xcc].insert(se << 8 | sh << 4 | cu);
  }
  int total = 0, rr = 0;
  for (int b = 0; b < nblk; ++b) rr += (int)((h[b * 2 + 1] & 0xf) == (unsigned)(b % 8));
  printf("%-34s %4d blocks in %7.3f ms, block b on XCC b %% 8: %5.1f %%:", name, nblk, ms, 100.0 * rr / nblk);
  for (auto& kv : per_xcc) { printf(" xcc%u:%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
  printf("  | distinct CUs %d\n", total);
  hipStreamDestroy(s);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("%s: %d CUs\n", p.name, p.multiProcessorCount);
  const int ncu = p.multiProcessorCount, words = (ncu + 31) / 32;
  unsigned* d;
  const int nblk = 2048;
  hipMalloc(&d, nblk * 8);
  auto mk = [&](auto pred) { std::vector<uint32_t> m(words, 0); for (int i = 0; i < ncu; ++i) if (pred(i)) m[i / 32] |= 1u << (i % 32); return m; };
  run("all", mk([](int) { return true; }), d, nblk);
  run("all (240 blocks)", mk([](int) { return true; }), d, 240);
  run("CUs 0..29 of every XCD (240 blocks)", mk([](int i) { return i / 8 < 30; }), d, 240);
  run("CUs 0..27 of every XCD (224 blocks)", mk([](int i) { return i / 8 < 28; }), d, 224);
  run("CUs 30..31 of every XCD (16 blocks)", mk([](int i) { return i / 8 >= 30; }), d, 16);
  run("first 32 bits", mk([](int i) { return i < 32; }), d, nblk);
  run("bits 0..223", mk([](int i) { return i < 224; }), d, nblk);
  run("bits 224..255", mk([](int i) { return i >= 224; }), d, nblk);
  run("i % 32 >= 28", mk([](int i) { return i % 32 >= 28; }), d, nblk);
  run("i % 32 < 28", mk([](int i) { return i % 32 < 28; }), d, nblk);
  run("i % 8 == 0", mk([](int i) { return i % 8 == 0; }), d, nblk);
  run("i / 8 % 8 == 7  (every 8th octet)", mk([](int i) { return (i / 8) % 8 == 7; }), d, nblk);
  run("i % 64 >= 56", mk([](int i) { return i % 64 >= 56; }), d, nblk);
  // two streams at once: complement masks, are they really disjoint in time and space?
  return 0;
}
