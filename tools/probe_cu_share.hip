// is FP64 MFMA (or VALU) throughput a per-SIMD or a per-CU resource?  k one-wave blocks per CU (k = 1..8; the dispatcher puts
// successive blocks of a CU on successive SIMDs, tools/probe_place.hip), each wave runs the same instruction loop; time per
// instruction by wall clock.      hipcc --offload-arch=gfx950 -O3 -o probe_cu_share probe_cu_share.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void __launch_bounds__(64) k_loop(double* out, int iters) {
  __shared__ double pad[2048];                      // 16 KB: at most 8-10 blocks per CU
  pad[threadIdx.x] = threadIdx.x;
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  d4 q0 = {0, 0, 0, 0}, q1 = q0;
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {          // 4 independent v_mfma_f64_4x4x4
      c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
    } else if (MODE == 1) {   // 2 independent v_mfma_f64_16x16x4
      q0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, q0, 0, 0, 0);
      q1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, q1, 0, 0, 0);
    } else if (MODE == 2) {   // 4 independent v_fma_f64
      c0 = fma(a, b, c0); c1 = fma(a, b, c1); c2 = fma(a, b, c2); c3 = fma(a, b, c3);
    } else {                  // dependent chain of v_mfma_f64_4x4x4 (the stepper's layer chains)
      c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, c0, c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, c0, c2, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, c0, c3, 0, 0, 0);
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = c0 + c1 + c2 + c3 + q0[0] + q1[1] + pad[(threadIdx.x + 1) & 63];
}
template <int MODE> void run(const char* name, int per_iter, double* d) {
  printf("%-44s", name);
  for (int k : {1, 2, 3, 4, 8}) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_loop<MODE>, dim3(256 * k), dim3(64), 0, 0, d, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_loop<MODE>, dim3(256 * k), dim3(64), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("  k=%d: %6.1f clk", k, ms * 1e-3 * 2.4e9 / ((double)iters * per_iter));
  }
  printf("\n");
}
int main() {
  double* d; hipMalloc(&d, 256 * 8 * 64 * sizeof(double));
  printf("clocks (at 2.4 GHz) per instruction per wave, k one-wave blocks per CU:\n");
  run<0>("v_mfma_f64_4x4x4, 4 independent", 4, d);
  run<3>("v_mfma_f64_4x4x4, dependent chain", 4, d);
  run<1>("v_mfma_f64_16x16x4, 2 independent", 2, d);
  run<2>("v_fma_f64, 4 independent", 4, d);
  return 0;
}
