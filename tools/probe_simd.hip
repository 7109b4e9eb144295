// Probe: on which SIMDs do the waves of a small workgroup land?  (HW_REG_HW_ID: wave_id[3:0] simd_id[5:4] cu_id[11:8] ...)
// build: hipcc --offload-arch=gfx950 -O3 -o probe_simd probe_simd.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out, int vg) {
  unsigned id = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));   // hwreg(HW_REG_HW_ID, 0, 32)
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = id;
  // keep the block alive a little so that blocks spread over the chip
  long long t0 = clock64();
  while (clock64() - t0 < 20000) {}
}
template <int REGS> __global__ void __launch_bounds__(128) kbig(unsigned* out, double* sink) {
  unsigned id = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 2 + threadIdx.x / 64] = id;
  double x[REGS];
#pragma unroll
  for (int i = 0; i < REGS; ++i) x[i] = threadIdx.x + i;
  for (int it = 0; it < 200; ++it) {
#pragma unroll
    for (int i = 0; i < REGS; ++i) x[i] = fma(x[i], 1.0000001, 0.5);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < REGS; ++i) s += x[i];
  sink[blockIdx.x * 128 + threadIdx.x] = s;
}
int main() {
  unsigned *d, h[4096];
  double* sink;
  hipMalloc(&d, sizeof(h)); hipMalloc(&sink, 8 * 128 * 1024);
  for (int wpb = 2; wpb <= 4; wpb *= 2) {
    k<<<256, 64 * wpb>>>(d, 0);
    hipMemcpy(h, d, 4 * 256 * wpb, hipMemcpyDeviceToHost);
    int same = 0;
    for (int b = 0; b < 256; ++b) {
      unsigned s0 = (h[b * wpb] >> 4) & 3, s1 = (h[b * wpb + 1] >> 4) & 3;
      if (s0 == s1) ++same;
    }
    printf("%d waves per block, small kernel: first two waves on the SAME simd in %d of 256 blocks; block 0 simds:", wpb, same);
    for (int w = 0; w < wpb; ++w) printf(" %u", (h[w] >> 4) & 3);
    printf("\n");
  }
  kbig<100><<<256, 128>>>(d, sink);
  hipMemcpy(h, d, 4 * 512, hipMemcpyDeviceToHost);
  int same = 0;
  for (int b = 0; b < 256; ++b) if (((h[2 * b] >> 4) & 3) == ((h[2 * b + 1] >> 4) & 3)) ++same;
  printf("2 waves per block, ~200-register kernel: same simd in %d of 256 blocks\n", same);
  kbig<150><<<256, 128>>>(d, sink);
  hipMemcpy(h, d, 4 * 512, hipMemcpyDeviceToHost);
  same = 0;
  for (int b = 0; b < 256; ++b) if (((h[2 * b] >> 4) & 3) == ((h[2 * b + 1] >> 4) & 3)) ++same;
  printf("2 waves per block, ~300-register kernel: same simd in %d of 256 blocks\n", same);
  return 0;
}
