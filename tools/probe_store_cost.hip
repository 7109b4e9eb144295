// What does a streamed store cost a LONE latency-chain wave (the stepper's forward pass: one wave per block, 1 wave per SIMD at
// most)?  Each iteration = one "layer": 9 v_mfma_f64_4x4x4 on three accumulators + 3 v_max_f64, then the layer's stores in one
// of several forms.  s_memtime around the loop, per wave; the launch is 256 one-wave blocks (one per CU) like k_ode_fwd at
// N = 4096.
//   hipcc --offload-arch=gfx950 -O3 -o probe_store_cost probe_store_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f64_4x4x4f64((a), (b), (c), 0, 0, 0)

// MODE 0: no stores | 1: 3 x b64 (all lanes) | 2: 2 x b64 + 1 x b64 under an exec branch (lanes < 32) | 3: 1 x b128 + 1 x b64
// 4: 2 x b64 + 1 x b64 with out-of-range offsets for lanes >= 32 (no branch) | 5: 3 x global_store (plain pointer) nt
// 6: 1 x b128 + 1 x b64 OOB-masked | 7: stores only every 2nd layer, 3 x b128 (two layers' registers paired)
template <int MODE>
__global__ void __launch_bounds__(64) k(double* __restrict__ out, unsigned long long* clk, int iters, double seed) {
  const int lane = threadIdx.x;
  double w[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) w[i] = seed * (i + 1) * 1e-3;
  double r0 = seed + lane * 1e-6, r1 = seed * 0.5, r2 = seed * 0.25;
  double p0 = 0, p1 = 0, p2 = 0;
  double* base = out + (long)blockIdx.x * ((long)iters * 192 + 1024);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    double a0 = 0.1, a1 = 0.2, a2 = 0.3;
    a0 = MFMA4(w[0], r0, a0); a1 = MFMA4(w[1], r0, a1); a2 = MFMA4(w[2], r0, a2);
    a0 = MFMA4(w[3], r1, a0); a1 = MFMA4(w[4], r1, a1); a2 = MFMA4(w[5], r1, a2);
    a0 = MFMA4(w[6], r2, a0); a1 = MFMA4(w[7], r2, a1); a2 = MFMA4(w[8], r2, a2);
    p0 = r0; p1 = r1; p2 = r2;
    r0 = __builtin_fmax(a0, 0.0) * 1e-3 + 0.5; r1 = __builtin_fmax(a1, 0.0) * 1e-3 + 0.5; r2 = __builtin_fmax(a2, 0.0) * 1e-3 + 0.5;
    double* A = base + (long)it * 192;      // 3 registers x 64 lanes per layer
    if (MODE == 0) continue;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(A, 0, 192 * 8, 0x00020000);
    const u2v x0 = {(unsigned)__double2loint(r0), (unsigned)__double2hiint(r0)};
    const u2v x1 = {(unsigned)__double2loint(r1), (unsigned)__double2hiint(r1)};
    const u2v x2 = {(unsigned)__double2loint(r2), (unsigned)__double2hiint(r2)};
    if (MODE == 1) {
      __builtin_amdgcn_raw_buffer_store_b64(x0, rs, lane * 8, 0, 2);
      __builtin_amdgcn_raw_buffer_store_b64(x1, rs, lane * 8, 512, 2);
      __builtin_amdgcn_raw_buffer_store_b64(x2, rs, lane * 8, 1024, 2);
    } else if (MODE == 2) {
      __builtin_amdgcn_raw_buffer_store_b64(x0, rs, lane * 8, 0, 2);
      __builtin_amdgcn_raw_buffer_store_b64(x1, rs, lane * 8, 512, 2);
      if (lane < 32) __builtin_amdgcn_raw_buffer_store_b64(x2, rs, lane * 8, 1024, 2);
    } else if (MODE == 3) {
      const u4v y = {x0[0], x0[1], x1[0], x1[1]};
      __builtin_amdgcn_raw_buffer_store_b128(y, rs, lane * 16, 0, 2);
      __builtin_amdgcn_raw_buffer_store_b64(x2, rs, lane * 8, 1024, 2);
    } else if (MODE == 4) {
      __builtin_amdgcn_raw_buffer_store_b64(x0, rs, lane * 8, 0, 2);
      __builtin_amdgcn_raw_buffer_store_b64(x1, rs, lane * 8, 512, 2);
      __builtin_amdgcn_raw_buffer_store_b64(x2, rs, lane < 32 ? lane * 8 : 0x7fffff00, 1024, 2);
    } else if (MODE == 5) {
      __builtin_nontemporal_store(r0, A + lane);
      __builtin_nontemporal_store(r1, A + 64 + lane);
      __builtin_nontemporal_store(r2, A + 128 + lane);
    } else if (MODE == 6) {
      const u4v y = {x0[0], x0[1], x1[0], x1[1]};
      __builtin_amdgcn_raw_buffer_store_b128(y, rs, lane * 16, 0, 2);
      __builtin_amdgcn_raw_buffer_store_b64(x2, rs, lane < 32 ? lane * 8 : 0x7fffff00, 1024, 2);
    } else if (MODE == 7) {
      if (it & 1) {
        const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(A - 192, 0, 384 * 8, 0x00020000);
        const u4v y0 = {(unsigned)__double2loint(p0), (unsigned)__double2hiint(p0), x0[0], x0[1]};
        const u4v y1 = {(unsigned)__double2loint(p1), (unsigned)__double2hiint(p1), x1[0], x1[1]};
        const u4v y2 = {(unsigned)__double2loint(p2), (unsigned)__double2hiint(p2), x2[0], x2[1]};
        __builtin_amdgcn_raw_buffer_store_b128(y0, rs2, lane * 16, 0, 2);
        __builtin_amdgcn_raw_buffer_store_b128(y1, rs2, lane * 16, 1024, 2);
        __builtin_amdgcn_raw_buffer_store_b128(y2, rs2, lane * 16, 2048, 2);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) clk[blockIdx.x] = t1 - t0;
  base[(long)iters * 192 + lane] = r0 + r1 + r2 + p0 + p1 + p2;
}
template <int MODE> void run(const char* name, double* out, unsigned long long* clk, int blocks) {
  const int iters = 31 * 16;
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(64), 0, 0, out, clk, 16, 1.0);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(64), 0, 0, out, clk, iters, 1.0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), clk, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  // s_memtime counts at 100 MHz on gfx950: report the event time per layer as well
  printf("%-64s blocks %4d: %7.1f us, %6.1f ns per layer, memtime median %llu\n", name, blocks, ms * 1e3, ms * 1e6 / iters, h[blocks / 2]);
}
int main() {
  const int maxb = 512;
  double* out; unsigned long long* clk;
  hipMalloc(&out, (size_t)maxb * (31 * 16 * 192 + 1024) * sizeof(double) + 4096); hipMalloc(&clk, maxb * sizeof(unsigned long long));
  for (int blocks : {256, 512}) {
    run<0>("no stores", out, clk, blocks);
    run<1>("3 x b64", out, clk, blocks);
    run<2>("2 x b64 + 1 x b64 under an exec branch", out, clk, blocks);
    run<4>("2 x b64 + 1 x b64 masked by an out-of-range offset", out, clk, blocks);
    run<3>("1 x b128 + 1 x b64", out, clk, blocks);
    run<6>("1 x b128 + 1 x b64 masked by offset", out, clk, blocks);
    run<5>("3 x global_store nt", out, clk, blocks);
    run<7>("every 2nd layer: 3 x b128 (two layers paired)", out, clk, blocks);
  }
  return 0;
}
