// Probe: FP64 execution resources on gfx950 (run on the GPU box; not part of the product library).
//   1. lane maps of v_mfma_f64_16x16x4_f64 (A, B, C/D) checked with asymmetric integer data
//   2. issue rate / dependent latency of v_fma_f64, v_mfma_f64_16x16x4_f64, v_mfma_f64_4x4x4_4b_f64
// build: hipcc --offload-arch=gfx950 -O3 -o probe_fp64 probe_fp64.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)

__global__ void k_layout(const double* A /*16x4*/, const double* B /*4x16*/, double* D /*16x16*/) {
  int l = threadIdx.x;
  double a = A[(l & 15) * 4 + (l >> 4)];   // A[i=l&15][k=l>>4]
  double b = B[(l >> 4) * 16 + (l & 15)];  // B[k=l>>4][j=l&15]
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];  // row=(l>>4)+4r, col=l&15
}

template <int MODE>
__global__ void k_rate(double* out, int iters, double seed) {
  int l = threadIdx.x;
  double a = seed + l * 1e-9, b = 1.0 - 1e-9 * l;
  long long t0, t1;
  if (MODE == 0) {  // 8 independent DFMA chains
    double x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
    t0 = clock64();
    for (int i = 0; i < iters; ++i) {
      x0 = fma(x0, b, a); x1 = fma(x1, b, a); x2 = fma(x2, b, a); x3 = fma(x3, b, a);
      x4 = fma(x4, b, a); x5 = fma(x5, b, a); x6 = fma(x6, b, a); x7 = fma(x7, b, a);
    }
    t1 = clock64();
    out[blockIdx.x * blockDim.x + l] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  } else if (MODE == 1) {  // 1 dependent DFMA chain
    double x0 = a;
    t0 = clock64();
    for (int i = 0; i < iters; ++i) {
      x0 = fma(x0, b, a); x0 = fma(x0, b, a); x0 = fma(x0, b, a); x0 = fma(x0, b, a);
      x0 = fma(x0, b, a); x0 = fma(x0, b, a); x0 = fma(x0, b, a); x0 = fma(x0, b, a);
    }
    t1 = clock64();
    out[blockIdx.x * blockDim.x + l] = x0;
  } else if (MODE == 2) {  // 4 independent MFMA 16x16x4 accumulators
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    t0 = clock64();
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    t1 = clock64();
    out[blockIdx.x * blockDim.x + l] = c0[0] + c1[1] + c2[2] + c3[3];
  } else if (MODE == 3) {  // dependent MFMA 16x16x4 chain (same accumulator)
    d4 c0 = {0, 0, 0, 0};
    t0 = clock64();
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    }
    t1 = clock64();
    out[blockIdx.x * blockDim.x + l] = c0[0] + c0[1] + c0[2] + c0[3];
  } else if (MODE == 4) {  // D of one MFMA feeds the B operand of the next (activation chain), 8 per iter
    d4 c0 = {a, a, a, a};
    t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        d4 z = {0, 0, 0, 0};
        z = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c0[q & 3], z, 0, 0, 0);
        c0 = z;
      }
    }
    t1 = clock64();
    out[blockIdx.x * blockDim.x + l] = c0[0] + c0[1] + c0[2] + c0[3];
  } else if (MODE == 5) {  // 4x4x4 (4 blocks), 8 independent accumulators
    double c[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) c[q] = 0;
    t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int q = 0; q < 8; ++q) c[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[q], 0, 0, 0);
    }
    t1 = clock64();
    double s = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += c[q];
    out[blockIdx.x * blockDim.x + l] = s;
  } else {  // MODE 6: dependent 4x4x4 chain
    double c = 0;
    t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int q = 0; q < 8; ++q) c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
    }
    t1 = clock64();
    out[blockIdx.x * blockDim.x + l] = c;
  }
  if (l == 0 && blockIdx.x == 0) out[gridDim.x * blockDim.x] = (double)(t1 - t0);
}

template <int MODE>
int run(const char* name, int blocks, int threads, int iters, double flop_per_iter_per_wave) {
  double* out;
  CK(hipMalloc(&out, sizeof(double) * (blocks * threads + 1)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k_rate<MODE><<<blocks, threads>>>(out, 10, 1.0);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  k_rate<MODE><<<blocks, threads>>>(out, iters, 1.0);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double cyc; CK(hipMemcpy(&cyc, out + blocks * threads, 8, hipMemcpyDeviceToHost));
  double waves = (double)blocks * threads / 64;
  printf("%-44s blocks=%5d thr=%4d  %8.3f ms  clock64/iter(8 ops)=%8.1f  -> %7.2f TFLOP/s\n", name, blocks, threads, ms,
         cyc / iters, waves * iters * flop_per_iter_per_wave / (ms * 1e-3) / 1e12);
  CK(hipFree(out));
  return 0;
}

int main() {
  // ---- 1. layout
  std::vector<double> A(64), B(64), D(256), R(256, 0.0);
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = 1 + i * 7 + k * 3;
  for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = 2 + k * 11 + j * 5 + (j * j) % 3;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
  double *dA, *dB, *dD;
  CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dD, 2048));
  CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
  k_layout<<<1, 64>>>(dA, dB, dD);
  CK(hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost));
  int bad = 0; for (int i = 0; i < 256; ++i) bad += (D[i] != R[i]);
  printf("mfma_f64_16x16x4 layout check (A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D row=(l>>4)+4r col=l&15): %s (%d mismatches)\n", bad ? "WRONG" : "OK", bad);
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s  CUs=%d  clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  // ---- 2. rates.  FLOP per iteration per wave: 8 ops x (64 lanes x 2) for DFMA; 8 x 2048 for 16x16x4; 8 x 512 for 4x4x4(4 blocks)
  const int it = 20000;
  run<0>("dfma 8 indep chains, 1 wave/SIMD", 256, 256, it, 8 * 128.0);
  run<0>("dfma 8 indep chains, 2 waves/SIMD", 256, 512, it, 8 * 128.0);
  run<0>("dfma 8 indep chains, 4 waves/SIMD", 256, 1024, it, 8 * 128.0);
  run<1>("dfma dependent chain, 1 wave/SIMD", 256, 256, it, 8 * 128.0);
  run<1>("dfma dependent chain, 4 waves/SIMD", 256, 1024, it, 8 * 128.0);
  run<1>("dfma dependent chain, 1 wave on chip", 1, 64, it, 8 * 128.0);
  run<2>("mfma16x16x4 4 indep acc, 1 wave/SIMD", 256, 256, it, 8 * 2048.0);
  run<2>("mfma16x16x4 4 indep acc, 2 waves/SIMD", 256, 512, it, 8 * 2048.0);
  run<3>("mfma16x16x4 dependent acc, 1 wave/SIMD", 256, 256, it, 8 * 2048.0);
  run<3>("mfma16x16x4 dependent acc, 2 waves/SIMD", 256, 512, it, 8 * 2048.0);
  run<3>("mfma16x16x4 dependent acc, 4 waves/SIMD", 256, 1024, it, 8 * 2048.0);
  run<3>("mfma16x16x4 dependent acc, 1 wave on chip", 1, 64, it, 8 * 2048.0);
  run<4>("mfma16x16x4 D->B chain, 1 wave/SIMD", 256, 256, it, 8 * 2048.0);
  run<4>("mfma16x16x4 D->B chain, 1 wave on chip", 1, 64, it, 8 * 2048.0);
  run<5>("mfma4x4x4 8 indep acc, 1 wave/SIMD", 256, 256, it, 8 * 512.0);
  run<6>("mfma4x4x4 dependent, 1 wave/SIMD", 256, 256, it, 8 * 512.0);
  run<6>("mfma4x4x4 dependent, 1 wave on chip", 1, 64, it, 8 * 512.0);
  return 0;
}
