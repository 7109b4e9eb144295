// xw_common.h -- shared device helpers for the gfx950 XNODE-WAN kernels.
//
// All dense work runs on v_mfma_f64_16x16x4_f64 (measured on MI355X: 64-cycle issue per SIMD = 77.6 TFLOP/s chip-wide,
// no extra latency on a chained accumulator, +17 cycles when a result feeds the next B operand;
// profiles/r01_probe_fp64.txt).  Lane maps of that instruction (checked by tools/probe_fp64.hip):
//     A[i][k]  : lane = i + 16 k          (i = lane & 15, k = lane >> 4),   one f64 per lane
//     B[k][j]  : lane = j + 16 k
//     D[i][j]  : lane = j + 16 (i & 3), register r = i >> 2       (row i = (lane >> 4) + 4 r, col j = lane & 15)
//
// "Chain layout": a [rows x 16 columns] activation tile kept as a d4 per lane, element (row = g + 4 r, col = n) with
// g = lane >> 4, n = lane & 15.  It is what an MFMA writes (D) and, register r taken as k-step r, exactly what the next
// MFMA reads as B (rows 4r + g) -- so a stack of small dense layers runs with no cross-lane data movement at all.
// The 16 columns are 16 Monte-Carlo paths (stepper) or 16 sample points (test network).
#pragma once
#include <hip/hip_runtime.h>

typedef double d4 __attribute__((ext_vector_type(4)));
#define XW_MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
// v_mfma_f64_4x4x4_4b_f64: four independent 4x4x4 products.  Lane maps (tools/probe_mfma4.hip):
//     A[blk][i][k] : lane = i + 4 blk + 16 k      B[blk][k][j] : lane = j + 4 blk + 16 k      D[blk][i][j] : lane = j + 4 blk + 16 i
// With the same A in all four blocks it is  D[4 rows x 16 columns] += A[4 x 4] B[4 x 16]: B is register k-block and D
// register row-block of the chain layout below, so it chains like the 16x16x4 form, at 4-row granularity (18 clocks).
#define XW_MFMA4(a, b, c) __builtin_amdgcn_mfma_f64_4x4x4f64((a), (b), (c), 0, 0, 0)

#define XW_E_DIMS (-1)
#define XW_ODE_MAX_LAYERS 10   /* u_layers: the depths xw_ode.hip instantiates (XW_ODE_DISPATCH); the ReLU-mask word of a stage holds 3 (m - 1) + 3 <= 32 bits.  Deeper fields (up to XWG_MAX_M): the generic path */
#define XW_E_ARG (-2)
#define XW_E_WORKSPACE (-3)
#define XW_E_COMM (-4)

// ---- parameter blob layouts (named_parameters() order of the reference modules) ---------------------------------
struct UOff {  // u_theta, src/model.py:78,85,130-138
  int IL0w, IL0b, IL2w, IL2b, IL4w, IL4b, Win, Winb, Wh, Whb, Wo, Wob, FLw, FLb, total, ldin;
};
__host__ __device__ inline UOff u_offsets(int d, int H, int K) {
  UOff o;
  int p = 0;
  o.IL0w = p; p += H;
  o.IL0b = p; p += H;
  o.IL2w = p; p += H * H;
  o.IL2b = p; p += H;
  o.IL4w = p; p += H * H;
  o.IL4b = p; p += H;
  o.ldin = d + 1 + H;
  o.Win = p; p += K * o.ldin;
  o.Winb = p; p += K;
  o.Wh = p; p += K * K;
  o.Whb = p; p += K;
  o.Wo = p; p += H * K;
  o.Wob = p; p += H;
  o.FLw = p; p += H;
  o.FLb = p; p += 1;
  o.total = p;
  return o;
}
struct VOff {  // v_phi, src/model.py:34-36
  int Vin, Vinb, Vh, Vhb, Vo, Vob, total, ldin;
};
__host__ __device__ inline VOff v_offsets(int d, int W) {
  VOff o;
  int p = 0;
  o.ldin = d + 1;
  o.Vin = p; p += W * o.ldin;
  o.Vinb = p; p += W;
  o.Vh = p; p += W * W;
  o.Vhb = p; p += W;
  o.Vo = p; p += W;
  o.Vob = p; p += 1;
  o.total = p;
  return o;
}

// ---- explicitly GLOBAL loads / stores ---------------------------------------------------------------------------------
// A pointer that went through an `asm volatile("" : "+s"(p))` (laundered, so that address arithmetic stays where it is
// written) has lost its address space: the compiler then emits FLAT instructions, which count on the LDS counter as well
// as on the vector-memory one -- every wait for an LDS read behind them becomes lgkmcnt(0) and drags the memory
// operation's issue into the LDS queue.  These casts say what the pointer is.
typedef const double __attribute__((address_space(1)))* xw_gcp;
typedef double __attribute__((address_space(1)))* xw_gp;
__device__ __forceinline__ double xw_ld_g(const double* p) { return *(xw_gcp)p; }
__device__ __forceinline__ void xw_st_g(double v, double* p) { *(xw_gp)p = v; }
__device__ __forceinline__ double xw_ld_nt(const double* p) { return __builtin_nontemporal_load((xw_gcp)p); }
__device__ __forceinline__ void xw_st_nt(double v, double* p) { __builtin_nontemporal_store(v, (xw_gp)p); }

// ---- lane helpers -------------------------------------------------------------------------------------------------
__device__ __forceinline__ int xw_lane() { return threadIdx.x & 63; }

// A-operand fragment of a row-major matrix Mx[rows x cols] (leading dimension ld): element (r0 + i, c0 + k), 0 outside
__device__ __forceinline__ double xw_fragA(const double* __restrict__ Mx, int ld, int rows, int cols, int r0, int c0) {
  const int l = xw_lane();
  const int r = r0 + (l & 15), c = c0 + (l >> 4);
  return (r < rows && c < cols) ? xw_ld_g(Mx + (r * ld + c)) : 0.0;
}
// A-operand fragment of the TRANSPOSE of Mx: element (r0 + i, c0 + k) of Mx^T, i.e. Mx[c0 + k][r0 + i]
__device__ __forceinline__ double xw_fragAT(const double* __restrict__ Mx, int ld, int rows, int cols, int r0, int c0) {
  const int l = xw_lane();
  const int r = r0 + (l & 15), c = c0 + (l >> 4);
  return (r < cols && c < rows) ? xw_ld_g(Mx + (c * ld + r)) : 0.0;
}
// the same with the lane id passed in (a laundered copy: keeps the address arithmetic where the caller wants it)
__device__ __forceinline__ double xw_fragAT_l(const double* __restrict__ Mx, int ld, int rows, int cols, int r0, int c0, int l) {
  const int r = r0 + (l & 15), c = c0 + (l >> 4);
  return (r < cols && c < rows) ? xw_ld_g(Mx + (c * ld + r)) : 0.0;
}
// A-operand of XW_MFMA4: the 4x4 block (rows r0.., columns c0..) of a row-major matrix, replicated over the lane blocks
__device__ __forceinline__ double xw_fragA4(const double* __restrict__ Mx, int ld, int rows, int cols, int r0, int c0) {
  const int l = xw_lane();
  const int r = r0 + (l & 3), c = c0 + (l >> 4);
  return (r < rows && c < cols) ? xw_ld_g(Mx + (r * ld + c)) : 0.0;
}
// the same block of the TRANSPOSE of Mx
__device__ __forceinline__ double xw_fragAT4(const double* __restrict__ Mx, int ld, int rows, int cols, int r0, int c0) {
  const int l = xw_lane();
  const int r = r0 + (l & 3), c = c0 + (l >> 4);
  return (r < cols && c < rows) ? xw_ld_g(Mx + (c * ld + r)) : 0.0;
}
// relu of one register: v_max_f64 (+ the compiler's canonicalising v_max in front of it; NaN -> 0 like `x > 0 ? x : 0`).
// NOT as inline asm: the hazard recogniser does not see into it and issues it right behind the MFMA that writes its
// input (6 wait states are required there) -- wrong results.
__device__ __forceinline__ double xw_relu1(double x) { return __builtin_fmax(x, 0.0); }
// chain-layout vector broadcast over the 16 columns: rows r0 + g + 4 r of b[rows]
__device__ __forceinline__ d4 xw_vecD(const double* __restrict__ b, int rows, int r0) {
  const int g = xw_lane() >> 4;
  d4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = r0 + g + 4 * r;
    v[r] = (i < rows) ? xw_ld_g(b + i) : 0.0;
  }
  return v;
}
// same, strided source (column of a row-major matrix)
__device__ __forceinline__ d4 xw_vecD_strided(const double* __restrict__ b, int stride, int rows, int r0) {
  const int g = xw_lane() >> 4;
  d4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = r0 + g + 4 * r;
    v[r] = (i < rows) ? xw_ld_g(b + (long)i * stride) : 0.0;
  }
  return v;
}
__device__ __forceinline__ d4 xw_relu(d4 z) {
  d4 o;
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = z[r] > 0.0 ? z[r] : 0.0;
  return o;
}
// x where b > 0, else +0, for b >= 0 (a ReLU output): the 64-bit pattern of b is non-zero exactly when b > 0, so the gate is
// a mask  -(min(lo | hi, 1))  and two 32-bit ANDs -- five 2.3-clock instructions where a compare and two v_cndmask_b32
// cost 4.6 + 2 x 6 clocks of the SIMD (profiles/r02_probe_coexec.txt)
__device__ __forceinline__ double xw_gate_pos(double b, double x) {
  const unsigned t = (unsigned)__double2loint(b) | (unsigned)__double2hiint(b);
  const int m = -(int)(t < 1u ? t : 1u);
  return __hiloint2double(__double2hiint(x) & m, __double2loint(x) & m);
}
__device__ __forceinline__ d4 xw_zero4() { d4 z = {0.0, 0.0, 0.0, 0.0}; return z; }

// tanh(x) = sign(x) (1 - e) / (1 + e), e = exp(-2|x|): 29 f64 VALU instructions (the library tanh has ~93; round 2's
// version of this one 38) -- in the stepper tanh is a fifth of the forward's vector instructions, and every one of them
// queues behind the test network's 64-clock MFMAs when the two share a SIMD.
//   |x| clamped to 40 (e < 2^-115);  n = rint(y log2 e) by the 1.5 * 2^52 shift (the integer then sits in the low word
//   of the shifted sum);  Cody-Waite reduction;  degree-12 Taylor polynomial on |r| <= ln2 / 2 (truncation 1.7e-16
//   relative to e <= 1);  2^n by an integer add into the exponent field (p is in [0.7, 1.42], n >= -116: always a normal
//   number);  division: v_rcp_f64 (2^-23) + one Newton step + one residual correction of the quotient.
// Max ABSOLUTE error 4e-16 over [-100, 100] (tests/test_host_logic.py emulates the sequence on the host).  NaN propagates
// (x * 0 is added to the result); so does an infinite x -- tanh(+-inf) = NaN here, where libm gives +-1: an infinite
// pre-activation only occurs in a run that has already diverged.
__device__ __forceinline__ double xw_tanh(double x) {
  const double am = fmin(fabs(x), 40.0);
  const double y = -2.0 * am;
  const double nb = fma(y, 1.4426950408889634, 6755399441055744.0);
  const double n = nb - 6755399441055744.0;
  double r = fma(n, -6.93147180369123816490e-01, y);
  r = fma(n, -1.90821492927058770002e-10, r);
  double p = 2.08767569878681e-09;
  p = fma(p, r, 2.505210838544172e-08);
  p = fma(p, r, 2.755731922398589e-07);
  p = fma(p, r, 2.7557319223985893e-06);
  p = fma(p, r, 2.48015873015873e-05);
  p = fma(p, r, 0.0001984126984126984);
  p = fma(p, r, 0.001388888888888889);
  p = fma(p, r, 0.008333333333333333);
  p = fma(p, r, 0.041666666666666664);
  p = fma(p, r, 0.16666666666666666);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  const double e = __hiloint2double(__double2hiint(p) + (__double2loint(nb) << 20), __double2loint(p));
  const double num = 1.0 - e, den = 1.0 + e;
  double rc = __builtin_amdgcn_rcp(den);
  rc = fma(fma(-den, rc, 1.0), rc, rc);
  double q = num * rc;
  q = fma(fma(-den, q, num), rc, q);
  return fma(x, 0.0, copysign(q, x));
}

// sum over the 4 lane groups g (lanes n, n+16, n+32, n+48): reduces the ROW index of a chain-layout partial
// gfx950 lane-swap instructions (VALU, no LDS round trip): v_permlane16_swap exchanges the odd 16-lane rows of one
// register with the even rows of the other, v_permlane32_swap the upper half with the lower half; fed the same value
// twice they return (x[row ^ 1] pairs) resp. (x[half ^ 1] pairs), so two swaps + two adds leave the total in every lane.
__device__ __forceinline__ double xw_sum_over_g(double x) {
  typedef unsigned xw_u2 __attribute__((ext_vector_type(2)));
  unsigned lo = __double2loint(x), hi = __double2hiint(x);
  xw_u2 a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  xw_u2 b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  x = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
  lo = __double2loint(x), hi = __double2hiint(x);
  a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
// Pairwise forms of the same swaps: reduce TWO values by one level with one swap (per 32-bit half) and one add.
//   xw_fold32(a, b): lanes 0..31 = a[l] + a[l + 32],  lanes 32..63 = b[l - 32] + b[l]
//   xw_fold16(a, b): row 0 = a.row0 + a.row1, row 1 = b.row0 + b.row1, row 2 = a.row2 + a.row3, row 3 = b.row2 + b.row3
// (rows = the four 16-lane groups g).  Three folds turn four per-group partials into one register that holds the four
// totals in the four lane groups -- 6 swaps + 3 adds where four xw_sum_over_g take 16 + 8.
__device__ __forceinline__ double xw_fold32(double a, double b) {
  typedef unsigned xw_u2 __attribute__((ext_vector_type(2)));
  const xw_u2 lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
  const xw_u2 hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
  return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ double xw_fold16(double a, double b) {
  typedef unsigned xw_u2 __attribute__((ext_vector_type(2)));
  const xw_u2 lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
  const xw_u2 hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
  return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
// sum over the 16 columns n (lanes within a 16-lane row)
// sum over the 16 lanes of a row (lane & 15), every lane gets the total.  The xor butterfly 1, 2, 4, 8 on DPP row operations
// instead of ds_bpermute (__shfl_xor): quad permutes for 1 and 2; once the quads are uniform, reversing a half row swaps its two
// quads (= xor 4) and reversing the row swaps its halves (= xor 8) -- the same additions in the same association, without the
// LDS crossbar's latency (4 x 2 dependent ds_bpermute per sum; the output layer of k_disc_rec takes 14 of these per tile).
#define XW_DPP_ADD(x, CTRL)                                                                                                    \
  {                                                                                                                            \
    const int lo_ = __builtin_amdgcn_update_dpp(0, __double2loint(x), (CTRL), 0xf, 0xf, true);                                 \
    const int hi_ = __builtin_amdgcn_update_dpp(0, __double2hiint(x), (CTRL), 0xf, 0xf, true);                                 \
    x += __hiloint2double(hi_, lo_);                                                                                           \
  }
__device__ __forceinline__ double xw_sum_over_n(double x) {
  XW_DPP_ADD(x, 0xB1)     // quad_perm [1, 0, 3, 2]
  XW_DPP_ADD(x, 0x4E)     // quad_perm [2, 3, 0, 1]
  XW_DPP_ADD(x, 0x141)    // row_half_mirror
  XW_DPP_ADD(x, 0x140)    // row_mirror
  return x;
}

// ---- transposition through LDS: chain layout -> operand layout for contractions over the 16 columns ---------------
// A [16 x 16] tile is stored as tile[row * 17 + col] (row stride 17 doubles keeps both phases nearly conflict-free).
// After xw_writeT, xw_readT(ks) returns, for lane (i = lane & 15, p = lane >> 4), element (row i, col 4 ks + p):
// the A operand of  D[i][j] += sum_col Q[i][col] * R[j][col]  and, read from R's tile, its B operand.
#define XW_TSTRIDE 17
#define XW_TTILE (16 * XW_TSTRIDE)
__device__ __forceinline__ void xw_writeT(double* tile, d4 q) {
  const int l = xw_lane();
  const int g = l >> 4, n = l & 15;
#pragma unroll
  for (int r = 0; r < 4; ++r) tile[(g + 4 * r) * XW_TSTRIDE + n] = q[r];
}
// the first NR registers only (rows g + 4 r, r < NR): tiles with fewer than 16 live rows.  The rows left alone keep
// stale LDS contents; as operand rows of an outer product they only reach accumulator rows / columns that are never stored.
template <int NR> __device__ __forceinline__ void xw_writeT_n(double* tile, d4 q) {
  const int l = xw_lane();
  const int g = l >> 4, n = l & 15;
#pragma unroll
  for (int r = 0; r < NR; ++r) tile[(g + 4 * r) * XW_TSTRIDE + n] = q[r];
}
// the same with the 16 columns stored at position 4 (n & 3) + (n >> 2): the tiles the duo sweep's chain wave posts for its
// partner, whose 4x4x4 operand (lane i + 4 b + 16 k = row i, path 4 k + b) then reads position 4 b + k -- a half wave covers
// positions {0,1,4,5,8,9,12,13} of four rows that sit 17 doubles apart: one bank conflict instead of twelve (read in path
// order, rows i and i + 2 overlap in six of their eight banks: 5.4 M conflict cycles per launch)
template <int NR> __device__ __forceinline__ void xw_writeT_pn(double* tile, d4 q) {
  const int l = xw_lane();
  const int g = l >> 4, n = l & 15, pn = ((n & 3) << 2) | (n >> 2);
#pragma unroll
  for (int r = 0; r < NR; ++r) tile[(g + 4 * r) * XW_TSTRIDE + pn] = q[r];
}
__device__ __forceinline__ double xw_readT(const double* tile, int ks) {
  const int l = xw_lane();
  return tile[(l & 15) * XW_TSTRIDE + 4 * ks + (l >> 4)];
}

// launch-error helper for the extern "C" wrappers
static inline int xw_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
