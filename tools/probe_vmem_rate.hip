// per-CU vector-memory issue cost: global loads of 8 B (dwordx2) and 16 B (dwordx4) per lane, fully coalesced (a wave reads
// 512 / 1024 contiguous bytes per instruction), from a buffer that stays in L2; k one-wave blocks per CU.
//   hipcc --offload-arch=gfx950 -O3 -o probe_vmem_rate probe_vmem_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int WIDE, int STRIDED>
__global__ void __launch_bounds__(64) k_ld(const double* __restrict__ src, double* out, int iters, long span) {
  __shared__ double pad[2048];
  pad[threadIdx.x] = 0;
  const int lane = threadIdx.x;
  // STRIDED: the four 16-lane groups read four 128-byte pieces 32 KB apart (the row-major activation store); else contiguous
  const long lane_off = STRIDED ? (long)(lane >> 4) * 4096 + (lane & 15) * (WIDE ? 2 : 1) : (long)lane * (WIDE ? 2 : 1);
  const double* p = src + ((long)blockIdx.x * 8192) % span + lane_off;
  double acc = 0;
  for (int i = 0; i < iters; ++i) {
    const double* q = p + ((long)i * 1024) % 65536;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (WIDE) {
        d2 v = *(const d2*)(q + u * 128);
        acc += v.x + v.y;
      } else {
        acc += q[u * 64];
      }
    }
  }
  out[blockIdx.x * 64 + lane] = acc + pad[(lane + 1) & 63];
}
template <int WIDE, int STRIDED> void run(const char* name, const double* src, double* out, long span) {
  printf("%-52s", name);
  for (int k : {1, 2, 4, 8}) {
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_ld<WIDE, STRIDED>), dim3(256 * k), dim3(64), 0, 0, src, out, 200, span);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_ld<WIDE, STRIDED>), dim3(256 * k), dim3(64), 0, 0, src, out, iters, span);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // clocks of the CU per load instruction: k waves x iters x 8 loads share the CU's memory path
    printf("  k=%d: %6.1f clk/instr/CU", k, ms * 1e-3 * 2.4e9 / ((double)iters * 8 * k));
  }
  printf("\n");
}
int main() {
  const long span = 1L << 22;   // 32 MB of doubles
  double *src, *out; hipMalloc(&src, (span + 131072) * sizeof(double)); hipMalloc(&out, 256 * 8 * 64 * sizeof(double));
  hipMemset(src, 0, (span + 131072) * sizeof(double));
  printf("clocks (2.4 GHz) of a CU per wave-wide load instruction, k one-wave blocks per CU:\n");
  run<0, 0>("8 B per lane, 512 contiguous bytes", src, out, span);
  run<1, 0>("16 B per lane, 1024 contiguous bytes", src, out, span);
  run<0, 1>("8 B per lane, four 128-byte pieces 32 KB apart", src, out, span);
  run<1, 1>("16 B per lane, four 256-byte pieces 32 KB apart", src, out, span);
  return 0;
}
