// Probe: can a wave's independent FP64 VALU work issue while its FP64 MFMA is in flight (gfx950)?
// per iteration: one v_mfma_f64_16x16x4_f64 on a dependent accumulator + NV independent v_fma_f64 (source-interleaved).
// build: hipcc --offload-arch=gfx950 -O3 -o probe_overlap probe_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NV, int NI>
__global__ void k(double* out, int iters, double seed) {
  int l = threadIdx.x;
  double a = seed + l * 1e-9, b = 1.0 - 1e-9 * l;
  d4 c = {0, 0, 0, 0};
  double x[8] = {a, a + 1, a + 2, a + 3, a + 4, a + 5, a + 6, a + 7};
  int y[8] = {l, l + 1, l + 2, l + 3, l + 4, l + 5, l + 6, l + 7};
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < NV; ++q) x[q & 7] = fma(x[q & 7], b, a);
#pragma unroll
      for (int q = 0; q < NI; ++q) y[q & 7] = y[q & 7] * 3 + l;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = clock64();
  double s = c[0] + c[1] + c[2] + c[3];
  for (int q = 0; q < 8; ++q) s += x[q] + y[q];
  out[blockIdx.x * blockDim.x + l] = s;
  if (l == 0 && blockIdx.x == 0) out[gridDim.x * blockDim.x] = (double)(t1 - t0) / (4.0 * iters);
}
template <int NV, int NI> void run(int threads) {
  double* out; hipMalloc(&out, 8 * (256 * threads + 1));
  k<NV, NI><<<256, threads>>>(out, 10, 1.0); hipDeviceSynchronize();
  k<NV, NI><<<256, threads>>>(out, 20000, 1.0); hipDeviceSynchronize();
  double cyc; hipMemcpy(&cyc, out + 256 * threads, 8, hipMemcpyDeviceToHost);
  printf("1 MFMA + %2d DFMA + %2d int-mad per group, %d wave(s)/SIMD: %7.1f clocks per group\n", NV, NI, threads / 256, cyc);
  hipFree(out);
}
int main() {
  run<0, 0>(256); run<2, 0>(256); run<4, 0>(256); run<6, 0>(256); run<8, 0>(256); run<12, 0>(256); run<16, 0>(256);
  run<0, 4>(256); run<0, 8>(256); run<0, 16>(256);
  run<0, 0>(512); run<4, 0>(512); run<8, 0>(512); run<16, 0>(512); run<0, 16>(512);
  return 0;
}
