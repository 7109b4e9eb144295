// Probe (GPU box only, not part of the product library): v_mfma_f64_4x4x4_4b_f64 as the building block of the stepper.
//   1. do the CBSZ / ABID fields broadcast ONE block's A operand to all four blocks on the f64 form?  (if yes, one
//      register carries four different 4x4 weight blocks and the instruction picks one: 4x fewer weight registers)
//   2. cycles of a "blocked" K x K layer (3 row blocks x 3 k blocks of 4x4x4, three independent accumulators, ReLU
//      between layers) against the same layer as 3 dependent 16x16x4 instructions
//   3. does a wave's own VALU work overlap its 4x4x4 MFMAs?
// build: hipcc --offload-arch=gfx950 -O3 -o probe_mfma4b probe_mfma4b.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double mfma4_b(double a, double b, double c, int abid) {   // (folds after unrolling)
  switch (abid & 3) {
    case 0: return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 2, 0, 0);
    case 1: return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 2, 1, 0);
    case 2: return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 2, 2, 0);
    default: return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 2, 3, 0);
  }
}
template <int ABID> __global__ void k_bcast(const double* a, const double* b, double* d) {
  int l = threadIdx.x;
  d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 2, ABID, 0);
}

// MODE 0: blocked layer, 9 x (4x4x4) per layer, 3 accumulators; 1: the same with cbsz broadcast operands;
// MODE 2: 3 x (16x16x4) dependent chain per layer (today's stepper);  8 layers per iteration
template <int MODE> __global__ void k_layer(double* out, int iters, double seed) {
  const int l = threadIdx.x;
  double w[9];
#pragma unroll
  for (int q = 0; q < 9; ++q) w[q] = 1e-3 * (seed + q) + 1e-6 * l;
  long long t0, t1;
  if (MODE < 2) {
    double z[3] = {seed + l, seed - l, seed * 0.5};
    t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int layer = 0; layer < 8; ++layer) {
        double r[3], n[3] = {0.1, 0.2, 0.3};
#pragma unroll
        for (int k = 0; k < 3; ++k) r[k] = fmax(z[k], 0.0);
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
          for (int rb = 0; rb < 3; ++rb)
            n[rb] = MODE == 0 ? __builtin_amdgcn_mfma_f64_4x4x4f64(w[rb * 3 + k], r[k], n[rb], 0, 0, 0)
                              : mfma4_b(w[(rb * 3 + k) >> 2], r[k], n[rb], rb * 3 + k);
#pragma unroll
        for (int k = 0; k < 3; ++k) z[k] = n[k];
      }
    }
    t1 = clock64();
    out[blockIdx.x * blockDim.x + l] = z[0] + z[1] + z[2];
  } else {
    d4 z = {seed + l, seed - l, seed * 0.5, 0.0};
    t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int layer = 0; layer < 8; ++layer) {
        d4 n = {0.1, 0.2, 0.3, 0.0};
#pragma unroll
        for (int k = 0; k < 3; ++k) n = __builtin_amdgcn_mfma_f64_16x16x4f64(w[k], fmax(z[k], 0.0), n, 0, 0, 0);
        z = n;
      }
    }
    t1 = clock64();
    out[blockIdx.x * blockDim.x + l] = z[0] + z[1] + z[2];
  }
  if (l == 0 && blockIdx.x == 0) out[gridDim.x * blockDim.x] = (double)(t1 - t0);
}

// 4 independent 4x4x4 + NV independent FMAs per group
template <int NV> __global__ void k_overlap(double* out, int iters, double seed) {
  const int l = threadIdx.x;
  double a = seed + 1e-9 * l, b = 1.0 - 1e-9 * l;
  double c[4] = {0, 0, 0, 0}, x[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) x[q] = a + q;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 4; ++q) c[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[q], 0, 0, 0);
#pragma unroll
    for (int q = 0; q < NV; ++q) x[q] = fma(x[q], b, a);
  }
  long long t1 = clock64();
  double s = c[0] + c[1] + c[2] + c[3];
#pragma unroll
  for (int q = 0; q < 16; ++q) s += x[q];
  out[blockIdx.x * blockDim.x + l] = s;
  if (l == 0 && blockIdx.x == 0) out[gridDim.x * blockDim.x] = (double)(t1 - t0);
}

__global__ void k_tput(double* out, int iters, double seed) {
  const int l = threadIdx.x;
  double a = seed + 1e-9 * (l & 63), b = 1.0 - 1e-9 * (l & 63);
  double c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 8; ++q) c[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[q], 0, 0, 0);
  }
  long long t1 = clock64();
  double s = 0;
#pragma unroll
  for (int q = 0; q < 8; ++q) s += c[q];
  out[blockIdx.x * blockDim.x + l] = s;
  if (l == 0 && blockIdx.x == 0) out[gridDim.x * blockDim.x] = (double)(t1 - t0);
}

template <typename F> double timed(F launch, double* out, int blocks, int threads) {
  launch(10);
  hipDeviceSynchronize();
  launch(2000);
  hipDeviceSynchronize();
  double clk;
  hipMemcpy(&clk, out + blocks * threads, 8, hipMemcpyDeviceToHost);
  return clk / 2000.0;
}

int main() {
  double ha[64], hb[64], hd[64], *da, *db, *dd;
  hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dd, 512);
  // A[blk][i][k] on lane i + 4 blk + 16 k holds 100 blk + 10 i + k + 1; B = identity per block (B[blk][k][j] = (k == j))
  // -> D[blk][i][j] (lane j + 4 blk + 16 i) = A[src blk][i][j]: prints which block's A every output block saw
  for (int l = 0; l < 64; ++l) {
    ha[l] = 100 * ((l >> 2) & 3) + 10 * (l & 3) + (l >> 4) + 1;
    hb[l] = ((l >> 4) == (l & 3)) ? 1.0 : 0.0;
  }
  hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
  for (int abid = 0; abid < 4; ++abid) {
    if (abid == 0) k_bcast<0><<<1, 64>>>(da, db, dd);
    if (abid == 1) k_bcast<1><<<1, 64>>>(da, db, dd);
    if (abid == 2) k_bcast<2><<<1, 64>>>(da, db, dd);
    if (abid == 3) k_bcast<3><<<1, 64>>>(da, db, dd);
    hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l) {
      const int j = l & 3, i = l >> 4;
      if (hd[l] != 100 * abid + 10 * i + j + 1) ok = 0;
    }
    printf("cbsz=2 abid=%d: every block sees A of block %d: %s   (block sources seen: %g %g %g %g)\n", abid, abid,
           ok ? "YES" : "no", (hd[0] - 1) / 100, (hd[4] - 1) / 100, (hd[8] - 1) / 100, (hd[12] - 1) / 100);
  }
  double* out;
  const int blocks = 256, threads = 256;
  hipMalloc(&out, 8 * (blocks * 1024 + 1));
  printf("K x K layer (K <= 12, 16 paths), clocks per layer, one wave per SIMD:\n");
  printf("  9 x mfma 4x4x4 (3 accumulators)        %7.1f\n", timed([&](int n) { k_layer<0><<<blocks, threads>>>(out, n, 1.0); }, out, blocks, threads) / 8);
  printf("  the same, cbsz=2 packed weight blocks   %7.1f\n", timed([&](int n) { k_layer<1><<<blocks, threads>>>(out, n, 1.0); }, out, blocks, threads) / 8);
  printf("  3 x mfma 16x16x4 (dependent)            %7.1f\n", timed([&](int n) { k_layer<2><<<blocks, threads>>>(out, n, 1.0); }, out, blocks, threads) / 8);
  printf("the same with two waves per SIMD:\n");
  printf("  9 x mfma 4x4x4                          %7.1f\n", timed([&](int n) { k_layer<0><<<blocks, 512>>>(out, n, 1.0); }, out, blocks, 512) / 8);
  printf("  3 x mfma 16x16x4                        %7.1f\n", timed([&](int n) { k_layer<2><<<blocks, 512>>>(out, n, 1.0); }, out, blocks, 512) / 8);
  printf("4 independent mfma 4x4x4 + k independent DFMA per group, one wave per SIMD (clocks per group):\n");
  printf("  k = 0  %6.1f\n", timed([&](int n) { k_overlap<0><<<blocks, threads>>>(out, n, 1.0); }, out, blocks, threads));
  printf("  k = 4  %6.1f\n", timed([&](int n) { k_overlap<4><<<blocks, threads>>>(out, n, 1.0); }, out, blocks, threads));
  printf("  k = 8  %6.1f\n", timed([&](int n) { k_overlap<8><<<blocks, threads>>>(out, n, 1.0); }, out, blocks, threads));
  printf("  k = 16 %6.1f\n", timed([&](int n) { k_overlap<16><<<blocks, threads>>>(out, n, 1.0); }, out, blocks, threads));
  printf("two waves per SIMD:\n");
  printf("  k = 0  %6.1f\n", timed([&](int n) { k_overlap<0><<<blocks, 512>>>(out, n, 1.0); }, out, blocks, 512));
  printf("  k = 8  %6.1f\n", timed([&](int n) { k_overlap<8><<<blocks, 512>>>(out, n, 1.0); }, out, blocks, 512));
  printf("  k = 16 %6.1f\n", timed([&](int n) { k_overlap<16><<<blocks, 512>>>(out, n, 1.0); }, out, blocks, 512));
  // 4. throughput of independent 4x4x4 MFMAs against waves per SIMD, by wall time (8 accumulators per wave)
  printf("independent mfma 4x4x4, 8 accumulators per wave, wall-clock throughput:\n");
  for (int wps = 1; wps <= 8; wps *= 2) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    k_tput<<<256, 256 * wps>>>(out, 10, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_tput<<<256, 256 * wps>>>(out, iters, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double clk;
    hipMemcpy(&clk, out + 256 * 256 * wps, 8, hipMemcpyDeviceToHost);
    const double flop = 512.0 * 8 * iters * 4.0 * wps * 256;
    printf("  %d wave(s)/SIMD: %8.3f ms  %7.1f clocks per 8 MFMAs per wave  -> %6.1f TFLOP/s  (%.2f clocks of one SIMD per MFMA)\n", wps, ms,
           clk / iters, flop / (ms * 1e-3) / 1e12, clk / iters / 8 / wps);
  }
  // 5. the layer kernels by wall time: 256-thread blocks, grid = 256 x (waves per SIMD)
  printf("K x K layer by wall time (ns per layer per wave; 2.4 clocks per ns), 256-thread blocks:\n");
  for (int wps = 1; wps <= 4; wps *= 2) {
    for (int mode = 0; mode < 3; mode += 2) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      const int iters = 4000;
      auto go = [&](int n) { if (mode == 0) k_layer<0><<<256 * wps, 256>>>(out, n, 1.0); else k_layer<2><<<256 * wps, 256>>>(out, n, 1.0); };
      go(10);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      go(iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("  %d wave(s)/SIMD  %s  %7.1f ns per layer  = %6.1f clocks  (%.1f clocks of the SIMD per layer-wave)\n", wps,
             mode == 0 ? "9 x 4x4x4  " : "3 x 16x16x4", ms * 1e6 / iters / 8, ms * 1e6 / iters / 8 * 2.4, ms * 1e6 / iters / 8 * 2.4 / wps);
    }
  }
  return 0;
}
