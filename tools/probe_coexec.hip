// Probe (wall-clock; clock64() under-counts with two waves per SIMD): do one wave's FP64 VALU instructions execute while
// ANOTHER wave of the same SIMD has an FP64 MFMA in flight?  Each wave: iters x { NM x v_mfma_f64_16x16x4 + NV x v_fma_f64 },
// all independent.  Reported: clocks of one SIMD per group = wall time x 2.4 GHz / iters / (waves per SIMD).
// build: hipcc --offload-arch=gfx950 -O3 -o probe_coexec probe_coexec.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NM, int NV, bool SMALL> __global__ void k(double* out, int iters, double seed) {
  const int l = threadIdx.x;
  double a = seed + 1e-9 * (l & 63), b = 1.0 - 1e-9 * (l & 63);
  d4 c[NM > 0 ? NM : 1];
  double cs[NM > 0 ? NM : 1];
  double x[NV > 0 ? NV : 1];
  for (int q = 0; q < (NM > 0 ? NM : 1); ++q) { c[q] = d4{0, 0, 0, 0}; cs[q] = 0; }
  for (int q = 0; q < (NV > 0 ? NV : 1); ++q) x[q] = a + q;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < NM; ++q) {
      if (SMALL) cs[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, cs[q], 0, 0, 0);
      else c[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[q], 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < NV; ++q) x[q] = fma(x[q], b, a);
  }
  double s = 0;
  for (int q = 0; q < (NM > 0 ? NM : 1); ++q) s += c[q][0] + c[q][3] + cs[q];
  for (int q = 0; q < (NV > 0 ? NV : 1); ++q) s += x[q];
  out[blockIdx.x * blockDim.x + l] = s;
}
// the same with NV 32-bit VALU instructions (KIND 0: v_add_u32 chain-free adds, 1: v_cndmask_b32 pairs via f64 select, 2: v_max_f64)
template <int NM, int NV, int KIND> __global__ void k32(double* out, int iters, double seed) {
  const int l = threadIdx.x;
  double a = seed + 1e-9 * (l & 63), b = 1.0 - 1e-9 * (l & 63);
  d4 c[NM > 0 ? NM : 1];
  unsigned x[NV > 0 ? NV : 1];
  double y[NV > 0 ? NV : 1];
  for (int q = 0; q < (NM > 0 ? NM : 1); ++q) c[q] = d4{0, 0, 0, 0};
  for (int q = 0; q < (NV > 0 ? NV : 1); ++q) { x[q] = l + q; y[q] = a - q; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < NM; ++q) c[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[q], 0, 0, 0);
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      if (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[q]) : "v"(l));
      if (KIND == 1) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[q]) : "v"(l));
      if (KIND == 2) asm volatile("v_max_f64 %0, %0, %1" : "+v"(y[q]) : "v"(b));
      if (KIND == 3) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(y[q]), "v"(b) : "vcc");
      if (KIND == 4) asm volatile("v_mov_b32 %0, %1" : "=v"(x[q]) : "v"(l));
    }
  }
  double s = 0;
  for (int q = 0; q < (NM > 0 ? NM : 1); ++q) s += c[q][0] + c[q][3];
  for (int q = 0; q < (NV > 0 ? NV : 1); ++q) s += x[q] + y[q];
  out[blockIdx.x * blockDim.x + l] = s;
}
template <int NM, int NV, int KIND> void run32(double* out, const char* what) {
  for (int wps = 1; wps <= 4; wps *= 2) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    k32<NM, NV, KIND><<<256 * wps, 256>>>(out, 100, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k32<NM, NV, KIND><<<256 * wps, 256>>>(out, iters, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("  %-34s %d wave(s)/SIMD: %8.1f clocks of the SIMD per group\n", what, wps, ms * 1e-3 * 2.4e9 / iters / wps);
  }
}
template <int NM, int NV, bool SMALL> void run(double* out, const char* what) {
  for (int wps = 1; wps <= 4; wps *= 2) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    k<NM, NV, SMALL><<<256 * wps, 256>>>(out, 100, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NM, NV, SMALL><<<256 * wps, 256>>>(out, iters, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("  %-34s %d wave(s)/SIMD: %8.1f clocks of the SIMD per group\n", what, wps, ms * 1e-3 * 2.4e9 / iters / wps);
  }
}
int main() {
  double* out;
  hipMalloc(&out, 8 * 256 * 4 * 256 + 64);
  printf("per group: NM mfma + NV v_fma_f64 (independent)\n");
  run<1, 0, false>(out, "1 mfma16x16x4");
  run<0, 8, false>(out, "8 dfma");
  run<1, 4, false>(out, "1 mfma16x16x4 + 4 dfma");
  run<1, 8, false>(out, "1 mfma16x16x4 + 8 dfma");
  run<1, 14, false>(out, "1 mfma16x16x4 + 14 dfma");
  run<2, 8, false>(out, "2 mfma16x16x4 + 8 dfma");
  run<4, 0, true>(out, "4 mfma4x4x4");
  run<4, 8, true>(out, "4 mfma4x4x4 + 8 dfma");
  run<4, 16, true>(out, "4 mfma4x4x4 + 16 dfma");
  run32<0, 16, 0>(out, "16 v_add_u32");
  run32<1, 16, 0>(out, "1 mfma16x16x4 + 16 v_add_u32");
  run32<1, 16, 1>(out, "1 mfma16x16x4 + 16 v_cndmask_b32");
  run32<1, 16, 4>(out, "1 mfma16x16x4 + 16 v_mov_b32");
  run32<1, 8, 2>(out, "1 mfma16x16x4 + 8 v_max_f64");
  run32<1, 8, 3>(out, "1 mfma16x16x4 + 8 v_cmp_lt_f64");
  return 0;
}
