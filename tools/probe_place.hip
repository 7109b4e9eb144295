// where do the waves of a 128-thread, 36 KB-LDS, 241-VGPR block land?  (XCC, SE, CU, SIMD per wave; blocks stay resident ~100 us)
//   hipcc --offload-arch=gfx950 -O3 -o probe_place probe_place.hip && ./probe_place
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
template <int WAVES, int SKIP>
__global__ void __launch_bounds__(64 * WAVES) k_place(unsigned* out, long spin) {
  __shared__ double lds[4488];                       // 35.9 KB like the duo sweep
  if (SKIP >= 0 && (int)(threadIdx.x >> 6) == SKIP) return;      // a wave that only shifts the placement of the next one
  lds[threadIdx.x & 127] = threadIdx.x;
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
  if ((threadIdx.x & 63) == 0) {
    const int slot = (SKIP >= 0 && (int)(threadIdx.x >> 6) > SKIP) ? (threadIdx.x >> 6) - 1 : (threadIdx.x >> 6);
    out[2 * (blockIdx.x * 2 + slot)] = hw;
    out[2 * (blockIdx.x * 2 + slot) + 1] = xcc;
  }
  if (lds[(threadIdx.x + 1) & 127] < 0) out[0] = 0;
}
int main() {
  unsigned* d;
  hipMalloc(&d, 2 * 2 * 2048 * sizeof(unsigned));
  for (int variant = 0; variant < 2; ++variant)
  for (int blocks : {256, 512, 768, 1024}) {
    hipMemset(d, 0, 2 * 2 * 2048 * sizeof(unsigned));
    if (variant == 0) hipLaunchKernelGGL((k_place<2, -1>), dim3(blocks), dim3(128), 0, 0, d, 10000L);   // 100 MHz wall clock: 100 us
    else hipLaunchKernelGGL((k_place<3, 1>), dim3(blocks), dim3(192), 0, 0, d, 10000L);                 // middle wave exits at once
    hipDeviceSynchronize();
    std::vector<unsigned> h(2 * 2 * blocks);
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::map<unsigned, int> perSimd, perCu;
    for (int w = 0; w < 2 * blocks; ++w) {
      unsigned hw = h[2 * w], xcc = h[2 * w + 1] & 0xf;
      unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      unsigned cuid = (xcc << 12) | (se << 8) | (sh << 4) | cu;
      perCu[cuid]++;
      perSimd[(cuid << 2) | simd]++;
    }
    int hs[9] = {0}, hc[17] = {0};
    for (auto& kv : perSimd) hs[kv.second < 8 ? kv.second : 8]++;
    for (auto& kv : perCu) hc[kv.second < 16 ? kv.second : 16]++;
    if (blocks == 512) {
      printf("   first CUs (xcc/se/sh/cu: simd of wave 0, wave 1 of each block there):");
      int shown = 0;
      for (auto& kv : perCu) {
        if (shown++ >= 3) break;
        printf("  [%03x:", kv.first);
        for (int w = 0; w < 2 * blocks; ++w) {
          unsigned hw = h[2 * w], xcc = h[2 * w + 1] & 0xf;
          unsigned cuid = (xcc << 12) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf);
          if (cuid == kv.first) printf(" %u", (hw >> 4) & 3);
        }
        printf("]");
      }
      printf("\n");
    }
    printf("%s %4d blocks (%4d waves): CUs used %3zu, SIMDs used %4zu | SIMDs with 1,2,3,4 waves: %d %d %d %d | CUs with 2,4,6,8 waves: %d %d %d %d\n",
           variant ? "3-wave blocks, middle wave exits:" : "2-wave blocks:", blocks, 2 * blocks, perCu.size(), perSimd.size(), hs[1], hs[2], hs[3], hs[4], hc[2], hc[4], hc[6], hc[8]);
  }
  return 0;
}
