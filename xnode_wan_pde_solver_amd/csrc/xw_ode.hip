// xw_ode.hip -- XNODE primal network u_theta on gfx950: fused fixed-grid ODE stepper, forward and reverse sweep.
//
// Replaces NeuralODE.forward + _F/_ODEField + torchdiffeq's Python stepping loop (src/model.py:87-112,140-156 of the
// reference) and the autograd replay of that loop (src/loss.py:55, src/training.py:137).
//
// One wave = 16 Monte-Carlo paths.  The hidden state y[H], the field's pre-activations z_j[K] and every cotangent live
// in registers in the "chain layout" of xw_common.h for the whole time loop; all weight matrices are MFMA A-fragments
// held in registers (13 + 14 f64 per lane at H=20, K=10).  Per field evaluation: 5 + 3(m-1) + 6 MFMAs forward and the
// same number for the vector-Jacobian product; parameter gradients are contractions over the 16 paths, done as 4-step
// MFMA outer products after an LDS transpose (xw_writeT / xw_readT), accumulated in registers over all stages, layers
// and time steps and written once per wave as a deterministic partial "slab".
//
// Memory traffic per path: x (d floats) once, the checkpoint y[L,H] written once / read once, u[L] -- everything else
// stays on chip.  At the headline size this kernel is bound by the FP64 matrix pipe, not by HBM.
#include <cstdlib>
#include <type_traits>
#include "xw_common.h"
#ifndef XW_ODE_FWD_WAVES
#define XW_ODE_FWD_WAVES 2     // waves per SIMD the forward pass's register allocation leaves room for (2: <= 256 registers; 3: <= 168)
#endif
#include "xnwan.h"

// One object per stepper width (Makefile: -DXW_ODE_H=.. -DXW_ODE_K=..), every depth m = 1..10 in it; the public entry
// points live in xw_ode_abi.hip and pick the object by (H, K).  Narrower networks run zero-padded inside the next larger
// width (exact: padding units stay identically zero, nets.Blob).
#if !defined(XW_ODE_H) || !defined(XW_ODE_K)
#error "compile with -DXW_ODE_H=<u_hidden_dim> -DXW_ODE_K=<u_hidden_hidden_dim>"
#endif
#define XW_CAT4_(a, b, c, d) a##b##c##d
#define XW_CAT4(a, b, c, d) XW_CAT4_(a, b, c, d)
#define XW_ODE_FN(name) XW_CAT4(name, XW_ODE_H, _, XW_ODE_K)
#ifdef XW_ODE_ONLY_M      /* development builds (ISA listings, A/B variants): one depth only */
#define XW_ODE_DISPATCH(CALL)                                    \
  switch (m) {                                                   \
    case XW_ODE_ONLY_M: { CALL(XW_ODE_H, XW_ODE_K, XW_ODE_ONLY_M) } \
    default: return XW_E_DIMS;                                   \
  }
#elif defined(XW_ODE_WIDE16)     /* (4 (m - 1) ReLU-mask bits per stage: one 32-bit word up to depth 9, two at depth 10 -- SaveX) */
#define XW_ODE_DISPATCH(CALL)                                    \
  switch (m) {                                                   \
    case 1: { CALL(XW_ODE_H, XW_ODE_K, 1) }                      \
    case 2: { CALL(XW_ODE_H, XW_ODE_K, 2) }                      \
    case 3: { CALL(XW_ODE_H, XW_ODE_K, 3) }                      \
    case 4: { CALL(XW_ODE_H, XW_ODE_K, 4) }                      \
    case 5: { CALL(XW_ODE_H, XW_ODE_K, 5) }                      \
    case 6: { CALL(XW_ODE_H, XW_ODE_K, 6) }                      \
    case 7: { CALL(XW_ODE_H, XW_ODE_K, 7) }                      \
    case 8: { CALL(XW_ODE_H, XW_ODE_K, 8) }                      \
    case 9: { CALL(XW_ODE_H, XW_ODE_K, 9) }                      \
    case 10: { CALL(XW_ODE_H, XW_ODE_K, 10) }                    \
    default: return XW_E_DIMS;                                   \
  }
#else
#define XW_ODE_DISPATCH(CALL)                                    \
  switch (m) {                                                   \
    case 1: { CALL(XW_ODE_H, XW_ODE_K, 1) }                      \
    case 2: { CALL(XW_ODE_H, XW_ODE_K, 2) }                      \
    case 3: { CALL(XW_ODE_H, XW_ODE_K, 3) }                      \
    case 4: { CALL(XW_ODE_H, XW_ODE_K, 4) }                      \
    case 5: { CALL(XW_ODE_H, XW_ODE_K, 5) }                      \
    case 6: { CALL(XW_ODE_H, XW_ODE_K, 6) }                      \
    case 7: { CALL(XW_ODE_H, XW_ODE_K, 7) }                      \
    case 8: { CALL(XW_ODE_H, XW_ODE_K, 8) }                      \
    case 9: { CALL(XW_ODE_H, XW_ODE_K, 9) }                      \
    case 10: { CALL(XW_ODE_H, XW_ODE_K, 10) }                    \
    default: return XW_E_DIMS;                                   \
  }
#endif

// The RECOMPUTING sweeps (no activation store: rk4, adjoint = True, XW_KEEP_ACT=0) are compiled into a second object per width
// (-DXW_ODE_PART_RECOMP): at (32, 12) the largest of them (midpoint / rk4 with weight gradients, the continuous adjoint) crash
// clang 22's 'AMDGPU Rewrite AGPR-Copy-MFMA' pass under -amdgpu-mfma-vgpr-form, which the kernels of the training path want
// (forward, sweeps from the store, narrow tiles: without it the (32, 12) object spilled thousands of registers to scratch).
// jobs: a BwdJobs, by address.
extern "C" int XW_ODE_FN(xw_ode_bwd_recomp_w)(const void* jobs, const double* t, const double* theta, int method, int L, int d,
                                              int m, int params, int adj, void* stream);

namespace {

// ---- explicit Runge-Kutta tableaux of the fixed-grid solvers (torchdiffeq fixed_grid: euler, midpoint, rk4 = 3/8 rule)
template <int METHOD> struct RK;
template <> struct RK<0> {
  static constexpr int S = 1;
  __device__ static constexpr double c(int) { return 0.0; }
  __device__ static constexpr double a(int, int) { return 0.0; }
  __device__ static constexpr double b(int) { return 1.0; }
};
template <> struct RK<1> {
  static constexpr int S = 2;
  __device__ static constexpr double c(int i) { return i == 1 ? 0.5 : 0.0; }
  __device__ static constexpr double a(int i, int j) { return (i == 1 && j == 0) ? 0.5 : 0.0; }
  __device__ static constexpr double b(int i) { return i == 1 ? 1.0 : 0.0; }
};
template <> struct RK<2> {
  static constexpr int S = 4;
  __device__ static constexpr double c(int i) { return i == 1 ? 1.0 / 3.0 : i == 2 ? 2.0 / 3.0 : i == 3 ? 1.0 : 0.0; }
  __device__ static constexpr double a(int i, int j) {
    return (i == 1 && j == 0) ? 1.0 / 3.0
         : (i == 2 && j == 0) ? -1.0 / 3.0
         : (i == 2 && j == 1) ? 1.0
         : (i == 3 && j == 0) ? 1.0
         : (i == 3 && j == 1) ? -1.0
         : (i == 3 && j == 2) ? 1.0 : 0.0;
  }
  __device__ static constexpr double b(int i) { return (i == 0 || i == 3) ? 0.125 : 0.375; }
};

template <int H, int K> struct Dim {
  static constexpr int HT = (H + 15) / 16;   // row tiles of an H-vector
  static constexpr int KSH = (H + 3) / 4;    // k-steps of a contraction over H
  static constexpr int KSK = (K + 3) / 4;    // k-steps of a contraction over K
  static constexpr int KR = (K + 1 + 3) / 4;   // live registers of a K-tile incl. the ones row (rows 0 .. K)
  static constexpr int KB = (K + 3) / 4;     // 4-row blocks of a K-vector (= live registers of its chain tile)
  static constexpr int HB = (H + 3) / 4;     // 4-row blocks of an H-vector
  __device__ static constexpr int HR(int ht) { return (H - 16 * ht) >= 16 ? 4 : (H - 16 * ht + 3) / 4; }        // rows of y
  __device__ static constexpr int HR1(int ct) { return (H + 1 - 16 * ct) >= 16 ? 4 : (H + 1 - 16 * ct + 3) / 4; }  // + time row
  static constexpr int CT = (H + 16) / 16;   // 16-row tiles of [y ; t]: the time row H is its own tile when H % 16 == 0
#ifndef XW_ODE_WIDE16
  static_assert(K <= 15, "row K of the 16-row K-tile is the ones row that collects the bias gradients");
  static_assert(HT <= 2 && CT <= 3, "H <= 32");
#else
  // the wide container (round 6, -DXW_ODE_WIDE16): whole 16-row tiles on v_mfma_f64_16x16x4 (the wide field family below), bias
  // gradients as row sums -- no ones row, no 4x4 blocks, no narrow tiles; a duo sweep of its own (duo_outer below)
  static_assert(K == 16 && H % 16 == 0 && H <= 64, "the wide container: K = 16, H a multiple of 16 up to 64");
#endif
};

#ifndef XW_ODE_WIDE16
// The field's layers run on v_mfma_f64_4x4x4_4b_f64: one instruction = a 4x4 weight block times 4 rows x 16 paths of
// the chain layout (its four "blocks" are the four groups of 4 paths; the weight block is replicated over them -- the
// CBSZ/ABID broadcast does nothing on the f64 form, profiles/r02_probe_mfma4b.txt).  A [K x K] layer is KB x KB = 9
// instructions of 18 clocks on KB INDEPENDENT accumulators instead of 3 dependent 16x16x4 instructions of 64 + 17 clocks
// on a tile with 10 of 16 rows live: 185 instead of 276 clocks per layer for the lone wave of a stepper tile, 16 % less
// matrix-pipe time when the chip is shared (same probe).
template <int H, int K> struct FieldW {      // forward operands: 4x4 blocks (row block, k block)
  double Wy[Dim<H, K>::KB][Dim<H, K>::HB];   // Win[:, d+1:]  [K x H]
  double Wh[Dim<H, K>::KB][Dim<H, K>::KB];   // Wh            [K x K]
  double Wo[Dim<H, K>::HB][Dim<H, K>::KB];   // Wo            [H x K]
  d4 wt, bh;                                 // Win[:, d] (time column), Wh.b
  d4 bo[Dim<H, K>::HT];                      // Wo.b
};
template <int H, int K> struct FieldWT {     // transposed operands for the vector-Jacobian product
  double WyT[Dim<H, K>::HB][Dim<H, K>::KB];  // [H x K]
  double WhT[Dim<H, K>::KB][Dim<H, K>::KB];  // [K x K]
  double WoT[Dim<H, K>::KB][Dim<H, K>::HB];  // [K x H]
};
#else
// ---- the wide container: the field on v_mfma_f64_16x16x4 -------------------------------------------------------------------
// K = 16 and H = 16 HT are whole 16-row tiles, so the 16x16x4 form wastes nothing (at K = 10 it ran 10 of 16 rows): a [K x K]
// layer is 4 chained instructions, Win's y-part H / 4, Wo HT x 4.  An A-fragment is ONE double per lane and (tile, k-step):
// 16 + 4 + 4 HT doubles hold the whole field (as 4x4 blocks replicated over the lane blocks it would be 144 doubles at (64, 16)).
template <int H, int K> struct FieldW {
  double Wy[Dim<H, K>::KSH];                   // Win[:, d+1:]  [K x H], k-steps over H
  double Wh[Dim<H, K>::KSK];                   // Wh            [K x K]
  double Wo[Dim<H, K>::HT][Dim<H, K>::KSK];    // Wo            [H x K], row tiles x k-steps over K
  d4 wt, bh;
  d4 bo[Dim<H, K>::HT];
};
template <int H, int K> struct FieldWT {
  double WoT[Dim<H, K>::KSH];                  // (Wo^T) [K x H]
  double WhT[Dim<H, K>::KSK];                  // (Wh^T) [K x K]
  double WyT[Dim<H, K>::HT][Dim<H, K>::KSK];   // (Wy^T) [H x K]
};
#endif
template <int M> struct Save {               // what the VJP of one field evaluation needs
  d4 z[M > 1 ? M - 1 : 1];                   // relu(z_0) .. relu(z_{m-2}): layer inputs; their sign pattern is the ReLU mask
  d4 a;                                      // tanh(z_{m-1})
  __device__ __forceinline__ bool pos(int j, int r) const { return z[j][r] > 0.0; }
  __device__ __forceinline__ double gate(int j, int r, double x) const { return z[j][r] > 0.0 ? x : 0.0; }   // relu'(z_j) * x
};
// the same for a sweep without weight gradients: the layer inputs are only needed as their ReLU masks, one bit per live
// register of the K-tile and layer in one word -- the x-only sweep then loads 1 word + tanh instead of m K doubles per stage.
// The forward pushes the bit (z > 0) of every pre-activation register into the word as it is produced (a compare and one
// add-with-carry: word = word + word + carry): (layer j, register r) was push number XW_KB j + r of XW_KB (M - 1) and sits
// at bit  XW_KB (M - 1) - 1 - (XW_KB j + r).  The test must be z > 0, not the sign bit: a layer whose units are all dead
// feeds EXACT zeros (+0, the biases start at zero) to the next one, and relu'(+0) = 0 in the reference (torch) -- with sign
// bits the boundary sweep of the d = 20 fixture was off by 5e-5.
#define XW_KB ((XW_ODE_K + 3) / 4)
// (more than 32 bits -- the wide container at depth 10: 36 -- take a second word, bits_hi = bits 32 and up; XW_MASK_WORDS)
#define XW_MASK_WORDS(M) ((XW_KB * ((M) - 1) > 32) ? 2 : 1)
template <int M> struct SaveX {
  static_assert(XW_KB * (M - 1) <= 64, "mask words");
  d4 a;
  unsigned bits, bits_hi;
  __device__ static constexpr int bit(int j, int r) { return XW_KB * (M - 1) - 1 - (XW_KB * j + r); }
  __device__ __forceinline__ bool pos(int j, int r) const { return ((bit(j, r) < 32 ? bits >> bit(j, r) : bits_hi >> (bit(j, r) - 32)) & 1u) != 0; }
  // relu'(z_j) * x as two 32-bit ANDs with the mask bit spread to 0 / ~0 (one v_bfe_i32 that does not depend on x):
  // a v_cndmask_b32 costs ~6 clocks of the SIMD, a v_and_b32 2.3 (profiles/r02_probe_coexec.txt), and the gate sits on the
  // adjoint chain's critical path once per layer and register
  __device__ __forceinline__ double gate(int j, int r, double x) const {
    const int m = bit(j, r) < 32 ? __builtin_amdgcn_sbfe((int)bits, bit(j, r), 1) : __builtin_amdgcn_sbfe((int)bits_hi, bit(j, r) - 32, 1);
    return __hiloint2double(__double2hiint(x) & m, __double2loint(x) & m);
  }
};

#ifndef XW_ODE_WIDE16
template <int H, int K>
__device__ __forceinline__ void load_field(const double* __restrict__ th, const UOff& o, int d, FieldW<H, K>& w) {
  typedef Dim<H, K> D;
  const double* Wy = th + o.Win + d + 1;
#pragma unroll
  for (int rb = 0; rb < D::KB; ++rb) {
#pragma unroll
    for (int kb = 0; kb < D::HB; ++kb) w.Wy[rb][kb] = xw_fragA4(Wy, o.ldin, K, H, 4 * rb, 4 * kb);
#pragma unroll
    for (int kb = 0; kb < D::KB; ++kb) w.Wh[rb][kb] = xw_fragA4(th + o.Wh, K, K, K, 4 * rb, 4 * kb);
  }
#pragma unroll
  for (int rb = 0; rb < D::HB; ++rb)
#pragma unroll
    for (int kb = 0; kb < D::KB; ++kb) w.Wo[rb][kb] = xw_fragA4(th + o.Wo, K, H, K, 4 * rb, 4 * kb);
#pragma unroll
  for (int ht = 0; ht < D::HT; ++ht) w.bo[ht] = xw_vecD(th + o.Wob, H, 16 * ht);
  w.wt = xw_vecD_strided(th + o.Win + d, o.ldin, K, 0);
  w.bh = xw_vecD(th + o.Whb, K, 0);
}
template <int H, int K>
__device__ __forceinline__ void load_field_T(const double* __restrict__ th, const UOff& o, int d, FieldWT<H, K>& w) {
  typedef Dim<H, K> D;
  const double* Wy = th + o.Win + d + 1;
#pragma unroll
  for (int rb = 0; rb < D::HB; ++rb)
#pragma unroll
    for (int kb = 0; kb < D::KB; ++kb) w.WyT[rb][kb] = xw_fragAT4(Wy, o.ldin, K, H, 4 * rb, 4 * kb);
#pragma unroll
  for (int rb = 0; rb < D::KB; ++rb) {
#pragma unroll
    for (int kb = 0; kb < D::KB; ++kb) w.WhT[rb][kb] = xw_fragAT4(th + o.Wh, K, K, K, 4 * rb, 4 * kb);
#pragma unroll
    for (int kb = 0; kb < D::HB; ++kb) w.WoT[rb][kb] = xw_fragAT4(th + o.Wo, K, H, K, 4 * rb, 4 * kb);
  }
}

#else
template <int H, int K>
__device__ __forceinline__ void load_field(const double* __restrict__ th, const UOff& o, int d, FieldW<H, K>& w) {
  typedef Dim<H, K> D;
  const double* Wy = th + o.Win + d + 1;
#pragma unroll
  for (int ks = 0; ks < D::KSH; ++ks) w.Wy[ks] = xw_fragA(Wy, o.ldin, K, H, 0, 4 * ks);
#pragma unroll
  for (int ks = 0; ks < D::KSK; ++ks) w.Wh[ks] = xw_fragA(th + o.Wh, K, K, K, 0, 4 * ks);
#pragma unroll
  for (int ht = 0; ht < D::HT; ++ht) {
#pragma unroll
    for (int ks = 0; ks < D::KSK; ++ks) w.Wo[ht][ks] = xw_fragA(th + o.Wo, K, H, K, 16 * ht, 4 * ks);
    w.bo[ht] = xw_vecD(th + o.Wob, H, 16 * ht);
  }
  w.wt = xw_vecD_strided(th + o.Win + d, o.ldin, K, 0);
  w.bh = xw_vecD(th + o.Whb, K, 0);
}
template <int H, int K>
__device__ __forceinline__ void load_field_T(const double* __restrict__ th, const UOff& o, int d, FieldWT<H, K>& w) {
  typedef Dim<H, K> D;
  const double* Wy = th + o.Win + d + 1;
#pragma unroll
  for (int ks = 0; ks < D::KSH; ++ks) w.WoT[ks] = xw_fragAT(th + o.Wo, K, H, K, 0, 4 * ks);          // (Wo^T)[i][4 ks + k] = Wo[4 ks + k][i]
#pragma unroll
  for (int ks = 0; ks < D::KSK; ++ks) w.WhT[ks] = xw_fragAT(th + o.Wh, K, K, K, 0, 4 * ks);
#pragma unroll
  for (int ht = 0; ht < D::HT; ++ht)
#pragma unroll
    for (int ks = 0; ks < D::KSK; ++ks) w.WyT[ht][ks] = xw_fragAT(Wy, o.ldin, K, H, 16 * ht, 4 * ks);   // (Wy^T)[16 ht + i][4 ks + k] = Wy[4 ks + k][16 ht + i]
}

#endif
// F([x, t, y]) of src/model.py:153-156: z0 = Win [x;t;y] + b (x part pre-contracted into xp), (m-1) tied ReLU layers,
// tanh, output layer.  y/out: HT chain tiles.
// Where the layer inputs of an evaluation go: nowhere, into registers (Save), or straight to the activation store.
struct SinkNone {
  __device__ __forceinline__ void fence() const {}
  __device__ __forceinline__ double relu(int, int, double z) const { return xw_relu1(z); }
  __device__ __forceinline__ void z(int, d4) const {}
  __device__ __forceinline__ void a(d4) const {}
};
template <int M> struct SinkSave {
  Save<M>& sv;
  __device__ __forceinline__ void fence() const {}
  __device__ __forceinline__ double relu(int, int, double z) const { return xw_relu1(z); }
  __device__ __forceinline__ void z(int j, d4 r) const { sv.z[j] = r; }
  __device__ __forceinline__ void a(d4 v) const { sv.a = v; }
};
#ifndef XW_ODE_WIDE16
template <int H, int K, int M, bool OUT = true, class Sink>
__device__ __forceinline__ void field_fwd(const FieldW<H, K>& w, double t, d4 xp, const d4 (&y)[Dim<H, K>::HT],
                                          d4 (&out)[Dim<H, K>::HT], const Sink& sink) {
  typedef Dim<H, K> D;
  // (k block outer, row block inner: consecutive instructions write different accumulators)
  d4 z = xw_zero4();
#pragma unroll
  for (int r = 0; r < D::KB; ++r) z[r] = fma(w.wt[r], t, xp[r]);
#pragma unroll
  for (int kb = 0; kb < D::HB; ++kb)
#pragma unroll
    for (int rb = 0; rb < D::KB; ++rb) z[rb] = XW_MFMA4(w.Wy[rb][kb], y[kb >> 2][kb & 3], z[rb]);
#pragma unroll
  for (int j = 0; j < M - 1; ++j) {
    d4 r = xw_zero4();
#pragma unroll
    for (int kb = 0; kb < D::KB; ++kb) r[kb] = sink.relu(j, kb, z[kb]);
    sink.fence();
    sink.z(j, r);
    d4 nz = w.bh;
#pragma unroll
    for (int kb = 0; kb < D::KB; ++kb)
#pragma unroll
      for (int rb = 0; rb < D::KB; ++rb) nz[rb] = XW_MFMA4(w.Wh[rb][kb], r[kb], nz[rb]);
    z = nz;
  }
  d4 a = xw_zero4();
#pragma unroll
  for (int kb = 0; kb < D::KB; ++kb) a[kb] = xw_tanh(z[kb]);
  sink.a(a);
  if (!OUT) return;
#pragma unroll
  for (int ht = 0; ht < D::HT; ++ht) out[ht] = w.bo[ht];
#pragma unroll
  for (int kb = 0; kb < D::KB; ++kb)
#pragma unroll
    for (int rb = 0; rb < D::HB; ++rb) out[rb >> 2][rb & 3] = XW_MFMA4(w.Wo[rb][kb], a[kb], out[rb >> 2][rb & 3]);
}

#else
template <int H, int K, int M, bool OUT = true, class Sink>
__device__ __forceinline__ void field_fwd(const FieldW<H, K>& w, double t, d4 xp, const d4 (&y)[Dim<H, K>::HT],
                                          d4 (&out)[Dim<H, K>::HT], const Sink& sink) {
  typedef Dim<H, K> D;
  d4 z;
#pragma unroll
  for (int r = 0; r < 4; ++r) z[r] = fma(w.wt[r], t, xp[r]);
#pragma unroll
  for (int ks = 0; ks < D::KSH; ++ks) z = XW_MFMA(w.Wy[ks], y[ks >> 2][ks & 3], z);
#pragma unroll
  for (int j = 0; j < M - 1; ++j) {
    d4 r;
#pragma unroll
    for (int kb = 0; kb < D::KB; ++kb) r[kb] = sink.relu(j, kb, z[kb]);
    sink.fence();
    sink.z(j, r);
    d4 nz = w.bh;
#pragma unroll
    for (int ks = 0; ks < D::KSK; ++ks) nz = XW_MFMA(w.Wh[ks], r[ks], nz);
    z = nz;
  }
  d4 a;
#pragma unroll
  for (int kb = 0; kb < D::KB; ++kb) a[kb] = xw_tanh(z[kb]);
  sink.a(a);
  if (!OUT) return;
#pragma unroll
  for (int ht = 0; ht < D::HT; ++ht) {
    out[ht] = w.bo[ht];
#pragma unroll
    for (int ks = 0; ks < D::KSK; ++ks) out[ht] = XW_MFMA(w.Wo[ht][ks], a[ks], out[ht]);
  }
}

#endif
// parameter-gradient accumulators of the field (chain-layout tiles of the gradient matrices)
#ifndef XW_ODE_WIDE16
template <int H, int K> struct FieldG {
  d4 Wh;                                          // rows K, cols K (+ column K = bias via a ones row)
  d4 Wy[Dim<H, K>::CT];                           // rows K, cols H (+ column H = time column)
  d4 Wo[Dim<H, K>::HT];                           // rows H, cols K (+ column K = bias)
};
#else
template <int H, int K> struct FieldG {           // wide container: no ones row / time row -- their gradients are elementwise sums
  d4 Wh;                                          // rows K, cols K
  d4 Wy[Dim<H, K>::CT];                           // rows K, cols H (tiles 0 .. HT-1; the last entry is not used)
  d4 Wo[Dim<H, K>::HT];                           // rows H, cols K
  d4 bh, wt;                                      // sum over evaluations of cot(z_{j+1}) (Wh.b) and of t cot(z_0) (Win's time column), per path
  d4 bo[Dim<H, K>::HT];                           // ... of cot(out) (Wo.b)
};
#endif

// D[i][j] += sum over the 16 paths of Q[i][path] * R[j][path]
// The block is exactly ONE wave and the LDS executes a wave's DS instructions in issue order, so the transposing
// write -> read round trip needs no s_barrier: a compiler-level fence keeps the program order of the accesses.
// QR / RR: live registers (4-row groups) of the two tiles
template <int QR = 4, int RR = 4>
__device__ __forceinline__ void outer_acc(d4& acc, d4 q, d4 r, double* lds) {
  xw_writeT_n<QR>(lds, q);
  xw_writeT_n<RR>(lds + XW_TTILE, r);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) acc = XW_MFMA(xw_readT(lds, ks), xw_readT(lds + XW_TTILE, ks), acc);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
// elementwise helpers on the first nr registers of a tile (nr is a compile-time constant after unrolling): an H-vector's
// last tile has H - 16 live rows, the rest is padding that no MFMA ever reads -- on this chip every FP64 VALU
// instruction of the lone sweep wave is time the matrix pipe stands still.
__device__ __forceinline__ void t_axpy(d4& y, double a, const d4& x, int nr) {
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (r < nr) y[r] += a * x[r];
}
__device__ __forceinline__ void t_scale(d4& y, double a, const d4& x, int nr) {
#pragma unroll
  for (int r = 0; r < 4; ++r) y[r] = r < nr ? a * x[r] : 0.0;
}
__device__ __forceinline__ void t_add(d4& y, const d4& x, int nr) {
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (r < nr) y[r] += x[r];
}
// Outer product whose R operand is a K-row activation followed by a row of ones (the bias gradient rides along as column
// K of the accumulator).  The ones row lives permanently in a dedicated LDS tile (written once per kernel): only the K
// real rows are stored, nothing is patched into registers.
template <int QR, int K>
__device__ __forceinline__ void outer_acc_ones(d4& acc, d4 q, d4 r, double* lds) {
  double* rt = lds + 2 * XW_TTILE;
  xw_writeT_n<QR>(lds, q);
  {
    const int l = xw_lane();
    const int g = l >> 4, n = l & 15;
#pragma unroll
    for (int rr = 0; rr < (K + 3) / 4; ++rr)
      if (4 * rr + 3 < K || g + 4 * rr < K) rt[(g + 4 * rr) * XW_TSTRIDE + n] = r[rr];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) acc = XW_MFMA(xw_readT(lds, ks), xw_readT(rt, ks), acc);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
// set chain-layout row `row` (0..15) of a tile to the value v in every column
__device__ __forceinline__ void set_row(d4& q, int row, double v) {
  if ((xw_lane() >> 4) == (row & 3)) q[row >> 2] = v;
}

// vector-Jacobian product of one field evaluation.  ob: cotangent of F's output; returns the cotangent of the y input
// in yb, adds the cotangent of z0 into xpb (= cotangent of the x-projection and of Win.b), and (PARAMS) accumulates the
// parameter gradients.
// The two halves of an outer product, so that the chain's next matrix instructions can be issued between the LDS
// stores and the loads that read them back transposed: the lone wave has nothing else to cover that round trip with.
template <int QR, int RR>
__device__ __forceinline__ void outer_post(d4 q, d4 r, double* lds) {                 // general R tile
  xw_writeT_n<QR>(lds, q);
  xw_writeT_n<RR>(lds + XW_TTILE, r);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_sched_barrier(0);
}
template <int QR, int K>
__device__ __forceinline__ void outer_post_ones(d4 q, d4 r, double* lds) {            // K rows + the permanent ones row
  double* rt = lds + 2 * XW_TTILE;
  xw_writeT_n<QR>(lds, q);
  const int l = xw_lane();
  const int g = l >> 4, n = l & 15;
#pragma unroll
  for (int rr = 0; rr < (K + 3) / 4; ++rr)
    if (4 * rr + 3 < K || g + 4 * rr < K) rt[(g + 4 * rr) * XW_TSTRIDE + n] = r[rr];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_sched_barrier(0);
}
// The operand loads of an outer product are issued right behind its stores (the LDS executes one wave's accesses in
// order) and BEFORE the chain's next matrix instructions, the four accumulating MFMAs after them: with the 18-clock
// 4x4x4 chain a layer's 9 instructions no longer cover a store -> load -> use round trip started behind them (the sweep
// with weight gradients stood at 209 us where the chain alone had gained 30 %).
struct OuterOps { double a[4], b[4]; };
__device__ __forceinline__ void outer_fetch(OuterOps& o, const double* qt, const double* rt) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    o.a[ks] = xw_readT(qt, ks);
    o.b[ks] = xw_readT(rt, ks);
  }
  __builtin_amdgcn_sched_barrier(0);
}
// operands of two products that share one side (two row tiles against one R tile, or one Q tile against two R tiles)
__device__ __forceinline__ void outer_fetch1(double (&x)[4], const double* t) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) x[ks] = xw_readT(t, ks);
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void outer_fire(d4& acc, const double (&a)[4], const double (&b)[4]) {
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) acc = XW_MFMA(a[ks], b[ks], acc);
  __builtin_amdgcn_sched_barrier(0);
}
// LDS plan of a sweep block (one wave): tiles of XW_TTILE doubles
//   0 Q | 1 R | 2 R of K rows + a permanent row of ones | 3 second Q (last H row tile) | 4, 5 further R tiles of [y ; t]
#define XW_SWEEP_TILES 6

// vector-Jacobian product of one field evaluation.  ob: cotangent of F's output; returns the cotangent of the y input
// in yb, adds the cotangent of z0 into xpb (= cotangent of the x-projection and of Win.b), and (PARAMS) accumulates the
// parameter gradients (outer products over the 16 paths; a row of ones / the time row in the R tile makes the bias and
// time-column gradients ride along as an extra accumulator column).
// OUTER: 0 = no weight gradients, 1 = this wave forms them itself (outer products through its own LDS tiles),
//        2 = "duo" sweep: this wave only posts the cotangent tiles (transposed) into `lds` = the evaluation's Q buffer
//            (DuoPlan), a partner wave of the block contracts them with the activations it loads itself.
#ifdef XW_ODE_WIDE16
// (wide container: every Q tile is a full 16-row tile of the 16x16x4 form -- cot(out) x HT, cot(z_{j+1}) for j = M-2 .. 0, cot(z_0))
template <int H, int K, int M> struct DuoPlan {
  static constexpr int HT = Dim<H, K>::HT;
  static constexpr int NQ = HT + M;
  __device__ static constexpr int off(int t) { return t * XW_TTILE; }
  static constexpr int BUF = NQ * XW_TTILE;
  static_assert(BUF >= 3 * XW_TTILE, "the chain wave's epilogue borrows a buffer for its three transpose tiles");
};
#else
template <int H, int K, int M> struct DuoPlan {
  static constexpr int HT = Dim<H, K>::HT;
  static constexpr int NQ = HT + (M - 1) + 1;     // Q tiles of one field evaluation: cot(out) x HT, cot(z_{j+1}) for j = M-2 .. 0, cot(z_0)
  // Tiles are packed by their LIVE rows (4-row groups): a partner reads 16 rows of every tile (xw_readT), the rows past a
  // tile's own are its successor's -- finite values that only reach accumulator rows which are never stored.  43 -> 16 KB
  // per buffer at (20, 10, 8): 4 instead of 2 resident sweep blocks per CU (the third job of a sub-step queued for LDS).
  static constexpr int HLAST = 4 * Dim<H, K>::HR(HT - 1);                   // rows of the last cot(out) tile
  static constexpr int KROWS = 4 * Dim<H, K>::KSK;                          // rows of a K-tile
  __device__ static constexpr int off(int t) {                             // first double of tile t
    return XW_TSTRIDE * (t < HT ? 16 * t : 16 * (HT - 1) + HLAST + KROWS * (t - HT));
  }
  static constexpr int BUF = XW_TSTRIDE * (16 * (HT - 1) + HLAST + KROWS * M + 16);   // (+ 16 rows: reads past the last tile)
  static_assert(BUF >= 3 * XW_TTILE, "the chain wave's epilogue borrows a buffer for its three transpose tiles");
};
#endif
#ifndef XW_ODE_WIDE16
template <int H, int K, int M, int OUTER, class SV>
__device__ __forceinline__ void field_vjp(const FieldW<H, K>& w, const FieldWT<H, K>& wT, double t, const SV& sv,
                                          const d4 (&yin)[Dim<H, K>::HT], const d4 (&ob)[Dim<H, K>::HT],
                                          d4 (&yb)[Dim<H, K>::HT], d4& xpb, FieldG<H, K>& G, double* lds) {
  typedef Dim<H, K> D;
  constexpr bool PARAMS = OUTER == 1;
  static_assert(D::HT <= 2, "two Q / R tile pairs in the LDS plan");
  const double* rt1 = lds + 2 * XW_TTILE;
  OuterOps o0;
  double q1[4];
  if (OUTER == 2) {
#pragma unroll
    for (int ht = 0; ht < D::HT; ++ht) {
      if (ht == 0) xw_writeT_pn<D::HR(0)>(lds, ob[0]);
      else xw_writeT_pn<D::HR(D::HT - 1)>(lds + DuoPlan<H, K, M>::off(ht), ob[ht]);
    }
  }
  if (PARAMS) {
    outer_post_ones<D::HR(0), K>(ob[0], sv.a, lds);
    if (D::HT > 1) {
      xw_writeT_n<D::HR(D::HT - 1)>(lds + 3 * XW_TTILE, ob[D::HT - 1]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    outer_fetch(o0, lds, rt1);
    if (D::HT > 1) outer_fetch1(q1, lds + 3 * XW_TTILE);
  }
  d4 ab = xw_zero4();
#pragma unroll
  for (int kb = 0; kb < D::HB; ++kb)
#pragma unroll
    for (int rb = 0; rb < D::KB; ++rb) ab[rb] = XW_MFMA4(wT.WoT[rb][kb], ob[kb >> 2][kb & 3], ab[rb]);
  if (PARAMS) {
    outer_fire(G.Wo[0], o0.a, o0.b);
    if (D::HT > 1) outer_fire(G.Wo[D::HT - 1], q1, o0.b);
  }
  d4 zb = xw_zero4();
#pragma unroll
  for (int r = 0; r < D::KSK; ++r) zb[r] = ab[r] * (1.0 - sv.a[r] * sv.a[r]);
#pragma unroll
  for (int j = M - 2; j >= 0; --j) {
    if constexpr (PARAMS) {
      outer_post_ones<D::KSK, K>(zb, sv.z[j], lds);
      outer_fetch(o0, lds, rt1);
    }
    if (OUTER == 2) xw_writeT_pn<D::KSK>(lds + DuoPlan<H, K, M>::off(D::HT + (M - 2 - j)), zb);
    d4 tt = xw_zero4();
#pragma unroll
    for (int kb = 0; kb < D::KB; ++kb)
#pragma unroll
      for (int rb = 0; rb < D::KB; ++rb) tt[rb] = XW_MFMA4(wT.WhT[rb][kb], zb[kb], tt[rb]);
    if (PARAMS) outer_fire(G.Wh, o0.a, o0.b);
#pragma unroll
    for (int r = 0; r < D::KSK; ++r) zb[r] = sv.gate(j, r, tt[r]);
  }
#pragma unroll
  for (int r = 0; r < D::KSK; ++r) xpb[r] += zb[r];
  if (OUTER == 2) xw_writeT_pn<D::KSK>(lds + DuoPlan<H, K, M>::off(D::HT + M - 1), zb);
  double rr[D::CT][4];
  if (PARAMS) {
    // one Q tile (the cotangent of z0) against the column tiles of [y ; t]: the time row makes column H collect the
    // time-column gradient (row H & 15 of tile H >> 4: a tile of its own when H is a multiple of 16)
    xw_writeT_n<D::KSK>(lds, zb);
#pragma unroll
    for (int ct = 0; ct < D::CT; ++ct) {
      d4 yy = ct < D::HT ? yin[ct < D::HT ? ct : 0] : xw_zero4();
      if (ct == (H >> 4)) set_row(yy, H & 15, t);
      double* rt = lds + (ct == 0 ? 1 : 3 + ct) * XW_TTILE;            // tiles 1, 4, 5
      if (ct == 0) xw_writeT_n<D::HR1(0)>(rt, yy);
      else if (ct == 1) xw_writeT_n<D::HR1(1)>(rt, yy);
      else xw_writeT_n<D::HR1(2)>(rt, yy);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_sched_barrier(0);
    outer_fetch1(o0.a, lds);
#pragma unroll
    for (int ct = 0; ct < D::CT; ++ct) outer_fetch1(rr[ct], lds + (ct == 0 ? 1 : 3 + ct) * XW_TTILE);
  }
#pragma unroll
  for (int ht = 0; ht < D::HT; ++ht) yb[ht] = xw_zero4();
#pragma unroll
  for (int kb = 0; kb < D::KB; ++kb)
#pragma unroll
    for (int rb = 0; rb < D::HB; ++rb) yb[rb >> 2][rb & 3] = XW_MFMA4(wT.WyT[rb][kb], zb[kb], yb[rb >> 2][rb & 3]);
  if (PARAMS) {
#pragma unroll
    for (int ct = 0; ct < D::CT; ++ct) outer_fire(G.Wy[ct], o0.a, rr[ct]);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

#else
template <int H, int K, int M, int OUTER, class SV>
__device__ __forceinline__ void field_vjp(const FieldW<H, K>& w, const FieldWT<H, K>& wT, double t, const SV& sv,
                                          const d4 (&yin)[Dim<H, K>::HT], const d4 (&ob)[Dim<H, K>::HT],
                                          d4 (&yb)[Dim<H, K>::HT], d4& xpb, FieldG<H, K>& G, double* lds) {
  typedef Dim<H, K> D;
  // OUTER: 0 = no weight gradients, 1 = this wave forms them itself (one LDS round trip per product, in the middle of the chain: the
  // recomputing sweeps), 2 = duo sweep: this wave only posts its cotangent tiles (transposed) into `lds` = the evaluation's Q buffer
  // (DuoPlan), the partner wave of the block (duo_outer) contracts them with the layer inputs it loads from the activation store
  constexpr bool PARAMS = OUTER == 1;
  constexpr bool POST = OUTER == 2;
  typedef DuoPlan<H, K, M> P;
  if (POST) {
#pragma unroll
    for (int ht = 0; ht < D::HT; ++ht) xw_writeT_n<4>(lds + P::off(ht), ob[ht]);
  }
  // cotangent of tanh(z_{m-1}): Wo^T cot(out), one chained accumulator over H / 4 k-steps
  d4 ab = xw_zero4();
#pragma unroll
  for (int ks = 0; ks < D::KSH; ++ks) ab = XW_MFMA(wT.WoT[ks], ob[ks >> 2][ks & 3], ab);
  if (PARAMS) {
    // dWo[16 ht ..][:] += cot(out)[ht] (x) tanh(z_{m-1}) over the 16 paths (LDS transposes, 4 k-steps each); dWo.b elementwise
#pragma unroll
    for (int ht = 0; ht < D::HT; ++ht) {
      outer_acc(G.Wo[ht], ob[ht], sv.a, lds);
      G.bo[ht] = G.bo[ht] + ob[ht];
    }
  }
  d4 zb;
#pragma unroll
  for (int r = 0; r < 4; ++r) zb[r] = ab[r] * (1.0 - sv.a[r] * sv.a[r]);
#pragma unroll
  for (int j = M - 2; j >= 0; --j) {
    if constexpr (PARAMS) {
      outer_acc(G.Wh, zb, sv.z[j], lds);
      G.bh = G.bh + zb;
    }
    if (POST) xw_writeT_n<4>(lds + P::off(D::HT + (M - 2 - j)), zb);      // cot(z_{j+1})
    d4 tt = xw_zero4();
#pragma unroll
    for (int ks = 0; ks < D::KSK; ++ks) tt = XW_MFMA(wT.WhT[ks], zb[ks], tt);
#pragma unroll
    for (int r = 0; r < 4; ++r) zb[r] = sv.gate(j, r, tt[r]);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) xpb[r] += zb[r];
  if (POST) xw_writeT_n<4>(lds + P::off(D::HT + M - 1), zb);              // cot(z_0)
  if (PARAMS) {
#pragma unroll
    for (int ct = 0; ct < D::HT; ++ct) outer_acc(G.Wy[ct], zb, yin[ct], lds);
#pragma unroll
    for (int r = 0; r < 4; ++r) G.wt[r] = fma(t, zb[r], G.wt[r]);
  }
#pragma unroll
  for (int ht = 0; ht < D::HT; ++ht) {
    yb[ht] = xw_zero4();
#pragma unroll
    for (int ks = 0; ks < D::KSK; ++ks) yb[ht] = XW_MFMA(wT.WyT[ht][ks], zb[ks], yb[ht]);
  }
}

#endif
// start scalar -> hidden state: initial_layers of src/model.py:78,97
template <int H, int K>
__device__ __forceinline__ void lift(const double* __restrict__ th, const UOff& o, double sv, d4 (&a0)[Dim<H, K>::HT],
                                     d4 (&a1)[Dim<H, K>::HT], d4 (&y)[Dim<H, K>::HT]) {
  typedef Dim<H, K> D;
#pragma unroll
  for (int ht = 0; ht < D::HT; ++ht)
    a0[ht] = xw_relu(xw_vecD(th + o.IL0w, H, 16 * ht) * sv + xw_vecD(th + o.IL0b, H, 16 * ht));
#pragma unroll
  for (int ht = 0; ht < D::HT; ++ht) {
    d4 v = xw_vecD(th + o.IL2b, H, 16 * ht);
#pragma unroll
    for (int ks = 0; ks < D::KSH; ++ks) v = XW_MFMA(xw_fragA(th + o.IL2w, H, H, H, 16 * ht, 4 * ks), a0[ks >> 2][ks & 3], v);
    a1[ht] = xw_relu(v);
  }
#pragma unroll
  for (int ht = 0; ht < D::HT; ++ht) {
    d4 v = xw_vecD(th + o.IL4b, H, 16 * ht);
#pragma unroll
    for (int ks = 0; ks < D::KSH; ++ks) v = XW_MFMA(xw_fragA(th + o.IL4w, H, H, H, 16 * ht, 4 * ks), a1[ks >> 2][ks & 3], v);
    y[ht] = v;
  }
}

// xp = Win.b + Win[:, :d] x   (time-invariant along a path: src/model.py:99,154)
template <int H, int K>
__device__ __forceinline__ d4 project_x(const double* __restrict__ th, const UOff& o, const double* __restrict__ xT,
                                        int N, int d, int ncl) {
  const int g = xw_lane() >> 4;
  d4 xp = xw_vecD(th + o.Winb, K, 0);
  for (int ks = 0; ks < (d + 3) / 4; ++ks) {
    const int i = 4 * ks + g;
    const double b = i < d ? xT[(long)i * N + ncl] : 0.0;
    xp = XW_MFMA(xw_fragA(th + o.Win, o.ldin, K, d, 0, 4 * ks), b, xp);
  }
  return xp;
}

// Up to XW_MAXJOBS independent groups of paths (e.g. interior + boundary sample, or two cotangents of the same sample)
// run in ONE launch: one wave per 16 paths only fills a quarter of the chip at N = 4096, and separate launches on
// separate streams do not reliably overlap (they can land on the same hardware queue).
#define XW_MAXJOBS 4
struct FwdJobs {
  const double* xT[XW_MAXJOBS];
  const double* start[XW_MAXJOBS];
  double* u[XW_MAXJOBS];
  double* Y[XW_MAXJOBS];
  double* act[XW_MAXJOBS];     // optional: stage activations of every step (ActLayout), read back by the sweeps
  int N[XW_MAXJOBS];
  int tile0[XW_MAXJOBS + 1];   // first block of each job
  int n;
  int x_only;                  // the stores of this launch only serve x-only sweeps (XwOdeFwdJob.act_x_only)
  int narrow;                  // narrow tiles (XwOdeFwdJob.narrow, xw_ode_n4.h)
  int prio;                    // wave priority (XW_ODE_PRIO - XwOdeFwdJob.prio_drop)
  double* zero16;              // optional: 16 doubles cleared by block 0 (the sub-step's partial-sum slots)
};
struct BwdJobs {
  const double* xT[XW_MAXJOBS];
  const double* start[XW_MAXJOBS];
  const double* Y[XW_MAXJOBS];
  const double* act[XW_MAXJOBS];
  const double* ubar[XW_MAXJOBS];
  double* gx[XW_MAXJOBS];
  double* gs[XW_MAXJOBS];
  double* gslab[XW_MAXJOBS];
  int N[XW_MAXJOBS];
  int tile0[XW_MAXJOBS + 1];
  int n;
  int x_ones;                  // gx, gs are those of the all-ones cotangent (ubar == 1 at every time index >= 1)
  int prio;                    // wave priority of this launch: XW_ODE_PRIO unless mode bits 5..6 lower it (XW_ODE_PRIO - bits)
  int spread;                  // duo sweep: spacer rounds of this many blocks between the rounds of tiles (k_ode_bwd_duo), 0 = none
  // cotangent from a residual (XwOdeBwdJob.res_*): ubar[l][n] = base + coef (res_u[l][n] - ref)
  const double* res_u[XW_MAXJOBS];
  const double* res_ref[XW_MAXJOBS];
  double res_coef[XW_MAXJOBS], res_base[XW_MAXJOBS];
  int res_first[XW_MAXJOBS];        // XwOdeBwdJob.res_first_only: 0 / 1 residual forms, 2 = the weak form's dI/du
  const double* res_w[XW_MAXJOBS];
  const double* res_c[XW_MAXJOBS];
  const double* res_cp[XW_MAXJOBS];
  double res_kappa2[XW_MAXJOBS];
  int res_wpp[XW_MAXJOBS];
};
// cotangent of u at (time index l, path col) of job `job`: a stored array, all ones, or formed from a residual on the fly
// (the initial-value and the boundary penalty: no cotangent kernel between the forward pass and these sweeps)
__device__ __forceinline__ double cot_u(const BwdJobs& jobs, int job, const double* __restrict__ ubar, int l, int N, int col,
                                        int L) {
  const double* __restrict__ ru = jobs.res_u[job];
  if (ru != nullptr) {
    if (jobs.res_first[job] == 2) {
      // dI/du of the weak form (xw_gen_cotangents' basis B, src/loss.py:64,70): coef d(c(u) u)/du v w, + base v at t_{L-1}
      const long p = (long)l * N + col;
      const double ul = ru[p], vl = jobs.res_ref[job][p];
      const double wl = jobs.res_wpp[job] ? jobs.res_w[job][p] : jobs.res_w[job][col];
      const double dcu = jobs.res_c[job] != nullptr ? jobs.res_c[job][p] + ul * jobs.res_cp[job][p] : jobs.res_kappa2[job] * ul;
      double gB = jobs.res_coef[job] * dcu * vl * wl;
      if (l == L - 1) gB += jobs.res_base[job] * vl;
      return gB;
    }
    double r = jobs.res_base[job];
    if (jobs.res_first[job]) {
      if (l == 0) r += jobs.res_coef[job] * (ru[col] - jobs.res_ref[job][col]);
    } else {
      r += jobs.res_coef[job] * (ru[(long)l * N + col] - jobs.res_ref[job][(long)l * N + col]);
    }
    return r;
  }
  return ubar != nullptr ? ubar[(long)l * N + col] : 1.0;
}
// ---- cotangent of u, one step ahead and without a branch around a load -------------------------------------------------
// cot_u() above branches on the kind of cotangent, and a value loaded inside a branch is used inside it: the
// wait in front of that use is vmcnt(0) -- it would drain the stage records this wave has just requested for the NEXT stage,
// i.e. expose a full memory latency in every step of a chain that is only ~2 - 4 k clocks long.  Here every kind is the same
// straight-line code: up to five loads from lane pointers prepared once (a pointer that a kind does not need aims at res_u /
// Y, finite data; its value is dropped by a select, never multiplied in), issued a whole step before their use.
//   affine kinds:  ub = cb + [st] + coef (ru - rf) [only at l = 0 when `first`]        (ones / stored / residual forms)
//   weak kind   :  ub = coef d(c u)/du v w  (+ base v at l = L - 1),  d(c u)/du = c + u c' (tabulated) or kappa2 u
struct Cot4 {
  const double *p0, *p1, *p2, *p3, *p4;      // st | ru, rf | (weak) u, v, w, c, c'      lane pointers at time index 0
  long s0, s1, s2;                           // strides (doubles) per time index of p0, (p1, p2 | p3, p4), p2 of the weak kind
  double cb, coef, base, kappa2;
  bool weak, use_st, use_res, first, tab, valid;
};
struct CotRaw { double a, b, c, d, e; };
__device__ __forceinline__ Cot4 make_cot(const BwdJobs& jobs, int job, int N, int col, bool valid, const double* dummy) {
  Cot4 c;
  const double* ru = jobs.res_u[job];
  const double* ubar = jobs.ubar[job];
  c.valid = valid;
  c.weak = ru != nullptr && jobs.res_first[job] == 2;
  c.use_res = ru != nullptr && !c.weak;
  c.use_st = ru == nullptr && ubar != nullptr;
  c.first = c.use_res && jobs.res_first[job] != 0;
  c.cb = ru != nullptr ? jobs.res_base[job] : (ubar != nullptr ? 0.0 : 1.0);
  c.coef = jobs.res_coef[job];
  c.base = jobs.res_base[job];
  c.kappa2 = jobs.res_kappa2[job];
  c.tab = c.weak && jobs.res_c[job] != nullptr;
  if (c.weak) {
    c.p0 = ru + col; c.p1 = jobs.res_ref[job] + col; c.p2 = jobs.res_w[job] + col;
    c.p3 = (c.tab ? jobs.res_c[job] : ru) + col; c.p4 = (c.tab ? jobs.res_cp[job] : ru) + col;
    c.s0 = N; c.s1 = N; c.s2 = jobs.res_wpp[job] ? N : 0;
  } else {
    c.p0 = (c.use_st ? ubar : dummy) + col; c.s0 = c.use_st ? N : 0;
    c.p1 = (c.use_res ? ru : dummy) + col; c.p2 = (c.use_res ? jobs.res_ref[job] : dummy) + col;
    c.s1 = (c.use_res && !c.first) ? N : 0; c.s2 = 0;
    c.p3 = c.p4 = dummy;
  }
  return c;
}
template <bool WEAK> __device__ __forceinline__ CotRaw cot_issue(const Cot4& c, int l) {
  CotRaw r;
  r.a = xw_ld_g(c.p0 + l * c.s0);
  r.b = xw_ld_g(c.p1 + l * c.s1);
  if (WEAK) {
    r.c = xw_ld_g(c.p2 + l * c.s2);
    r.d = xw_ld_g(c.p3 + l * c.s1);
    r.e = xw_ld_g(c.p4 + l * c.s1);
  } else {
    r.c = xw_ld_g(c.p2 + l * c.s1);
    r.d = r.e = 0.0;
  }
  return r;
}
template <bool WEAK> __device__ __forceinline__ double cot_value(const Cot4& c, const CotRaw& r, int l, int L) {
  double ub;
  if (WEAK) {
    const double dcu = c.tab ? fma(r.a, r.e, r.d) : c.kappa2 * r.a;        // (as cot_u: c + u c', or kappa2 u)
    ub = c.coef * dcu * r.b * r.c;
    if (l == L - 1) ub = fma(c.base, r.b, ub);
  } else {
    ub = c.cb;
    if (c.use_st) ub += r.a;
    if (c.use_res && (!c.first || l == 0)) ub = fma(c.coef, r.b - r.c, ub);
  }
  return c.valid ? ub : 0.0;
}

// vb: the 16-path tile of the launch this wave works on (= blockIdx.x in the one-tile-per-block kernels)
template <typename J> __device__ __forceinline__ int find_job(const J& jobs, int vb) {
  int j = 0;
#pragma unroll
  for (int k = 1; k < XW_MAXJOBS; ++k)
    if (k < jobs.n && vb >= jobs.tile0[k]) j = k;
  return j;
}

// The stepper's waves are latency chains that share their SIMD with a wave of the test network in the first phase of a
// sub-step; raised wave priority lets them issue whenever they are ready (the throughput-bound neighbour fills the rest):
// 0.832 -> 0.808 ms per discriminator sub-step, 0.574 -> 0.567 per generator sub-step.
#define XW_ODE_PRIO 3
__device__ __forceinline__ void xw_setprio(int p) {       // (s_setprio takes an immediate; p is uniform over the launch)
  if (p >= 3) __builtin_amdgcn_s_setprio(3);
  else if (p == 2) __builtin_amdgcn_s_setprio(2);
  else if (p == 1) __builtin_amdgcn_s_setprio(1);
}

// ---- stage activations kept from the forward pass ---------------------------------------------------------------------
// The sweeps need, for every stage of every step, the layer inputs relu(z_j), the tanh output and the stage input.
// Recomputing them from the checkpoint y_l costs the lone sweep wave 58 MFMAs and ~280 FP64 VALU instructions per step
// (two tanh blocks); HBM has 288 GB and is idle on this path, so the forward can simply store them:
//   act[l][tile of 16 paths][row][16],  l = 0 .. L-2,  rows: stage i -> i * M K + j K + k  (j < M-1: relu(z_j),
//   j = M-1: tanh(z_{M-1})), then the inputs of the stages i >= 1: S M K + (i-1) H + h.  Tile-major: what a wave stores
//   or loads for one step is ONE contiguous stretch (23 KB at 180 rows); path-major rows made it 180 pieces of 128 bytes,
//   32 KB apart, and the x-only sweep (183 MB in 0.1 ms) ran at the DRAM efficiency of that pattern.
//   Inside a tile every block of p <= 4 rows that one register of a chain tile covers is stored PATH-major,
//   [16 paths][p rows]: the partner wave of the duo sweep needs such a block as the B operand of a 4x4x4 MFMA -- lane
//   j + 4 b + 16 k = (row j, path 4 k + b) -- which is then lane-linear, 512 contiguous bytes.  Row-major, the four lanes of a
//   quad sat in four different 128-byte lines and the CU's address unit was stalled by the L1 a quarter of the launch
//   with three sweep tiles on a CU (TA_ADDR_STALLED_BY_TC 24.8 M -> 7.7 M cycles per three-job launch, average vector-memory
//   latency per instruction 29 -> 17 units; three concurrent jobs 170 -> 163 us, 16384 paths 244 -> 215 us, generator
//   sub-step 0.518 -> 0.510 ms).  What still separates two tiles of a CU (102 us alone, 148 us each) is the amount of
//   read data in flight per CU: 12.7 KB per field evaluation and tile at ~570 clocks of L2 latency.
// 180 doubles per path and step at (H, K, m) = (20, 10, 8), midpoint: 183 MB for 4096 paths x 32 times.
template <int H, int K, int M, int S> struct ActLayout {
  static constexpr int STAGE = M * K;
  static constexpr int YI = S * STAGE;
  static constexpr int ROWS = S * STAGE + (S - 1) * H;
  static constexpr int MASK = ROWS;                      // + 2 rows (64 words) per stage and mask word: the ReLU masks of SaveX
  static constexpr int MR = 2 * XW_MASK_WORDS(M);
  static constexpr int TOTAL = ROWS + MR * S;
};
// rows [row0, row0 + nrows) of the record <-> the first registers of a chain tile (row g + 4 r).  Addresses are
// formed as  (uniform row pointer) + (32-bit lane offset)  so that they cost SGPRs, not a VGPR pair per stored row.
struct ActLane {       // (unsigned: uniform pointer + zero-extended 32-bit lane offset is the scalar-base addressing form of the
                       //  global loads / stores -- with signed offsets every 4 KB of the record cost a 64-bit vector add)
  unsigned off;        // BYTES inside a full block of four rows: 8 (4 * column + g)
  unsigned off_part;   // inside the last, partially filled block of a K-row tile (p = K mod 4 rows): 8 (p * column + min(g, p - 1))
  unsigned off_part_st;   // the same for STORES: lanes without a row of their own (g >= p) get an offset beyond every record --
                          // the buffer's range check drops their store.  As a branch on the lane group (s_and_saveexec /
                          // s_cbranch_execz / s_or exec around one store per layer) the partial block cost the lone forward wave
                          // ~56 clocks per layer, 2.5 x the three stores themselves (tools/probe_store_cost.hip,
                          // profiles/r06_probe_store_cost.txt), and cut the time loop into a basic block per layer.
  bool valid;
};
#define XW_ACT_OOB 0x7fffff00u
__device__ __forceinline__ ActLane act_lane(int N, int col, bool valid, int krows) {
  const int g = xw_lane() >> 4, p = krows & 3;
  const int gp = p ? (g < p ? g : p - 1) : g;
  const unsigned part = 8u * (unsigned)((p ? p : 4) * col + gp);
  return ActLane{8u * (unsigned)(4 * col + g), part, (p == 0 || g < p) ? part : XW_ACT_OOB, valid};
}
// The forward pass writes the record through a BUFFER descriptor of the step's tile (base = the tile's 23 KB, built from
// scalars once per step): a store is then  descriptor + scalar row offset + 32-bit lane offset  and costs no vector
// instruction for its address.  With plain global stores the compiler folds all row offsets of a step onto one 64-bit
// vector address and re-bases it every 4 KB: 58 v_add_co / v_addc per step, each of which queues for the shared pipe.
typedef unsigned xw_u2v __attribute__((ext_vector_type(2)));
struct ActBuf {
  __amdgpu_buffer_rsrc_t rsrc;
};
__device__ __forceinline__ ActBuf act_buf(double* A, int bytes) {
  return ActBuf{__builtin_amdgcn_make_buffer_rsrc(A, 0, bytes, 0x00020000)};
}
__device__ __forceinline__ void act_store(const ActBuf& B, int row0, int nrows, int N, const ActLane& q, d4 v) {
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (4 * r < nrows) {
      const int so = (row0 + 4 * r) * N * 8;                                // uniform: the block of rows 4 r .. 4 r + 3
      // (the element goes through a scalar first: __builtin_bit_cast applied to v[r] directly reads element 0 of the vector)
      const double x = v[r];
      const xw_u2v w2 = {(unsigned)__double2loint(x), (unsigned)__double2hiint(x)};
      // (nt: streamed, read once, by a sweep; no branch: the lanes past a partial block's rows carry an out-of-range offset.
      //  Every caller's lanes are valid -- padding paths of the last tile hold finite copies and have slots of their own)
      __builtin_amdgcn_raw_buffer_store_b64(w2, B.rsrc, (int)(4 * r + 4 <= nrows ? q.off : q.off_part_st), so, 2);
    }
}
__device__ __forceinline__ d4 act_load(const double* __restrict__ A, int row0, int nrows, int N, const ActLane& q) {
  d4 v = xw_zero4();
  const char* base = reinterpret_cast<const char*>(A + (long)row0 * N);
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (4 * r < nrows) {
      const char* __restrict__ blk = base + (long)(4 * r) * N * 8;         // uniform
      v[r] = __builtin_nontemporal_load(reinterpret_cast<const double*>(blk + (4 * r + 4 <= nrows ? q.off : q.off_part)));   // padding rows: any finite value
    }
  return v;
}

// streams the layer inputs of one stage into the activation store as they are produced (no register copy kept)
// FULL = false: only what a sweep WITHOUT weight gradients reads back (the tanh rows and the mask words)
template <int K, int M, bool FULL = true> struct SinkAct {
  const ActBuf& A;             // record of this step
  int row0, N;
  const ActLane& q;
  unsigned& bits;              // (z > 0) of the stage's pre-activations, pushed in (layer, register) order (SaveX)
  unsigned& bits_hi;           // (the second word: only where XW_MASK_WORDS(M) == 2)
  // relu of one pre-activation register: its bit (z > 0) goes into the mask word (v_cmp_gt_f64 + v_addc_co_u32), then
  // ONE v_max_f64 (3 vector instructions per register where compare, select, shift-or and two v_max_f64 were 4.5).  As a builtin the maximum comes with a canonicalising v_max_f64 z, z in front of it (45 extra vector
  // instructions per step); as inline assembly it would escape the hazard recogniser, which must keep a VALU read 6+ wait
  // states behind the MFMA that wrote z -- so the assembly takes the mask word as an (unused) operand: it then follows the
  // compare, a visible read of the same MFMA result for which the wait states have been inserted.
  __device__ __forceinline__ double relu(int, int kb, double z) const {
    const bool kb_last = kb == (K + 3) / 4 - 1;
    // (the compare is compiler-visible -- it is the read of z the hazard recogniser sees --, its lane mask goes into the
    //  assembly as a scalar pair and becomes the carry-in of  word = word + word + carry;  written in C the compiler
    //  turns the add-with-carry back into compare + select + shift-or)
    const unsigned long long open = __builtin_amdgcn_ballot_w64(z > 0.0);
    double r;
    // (a matrix instruction may not read the maximum in the two issue slots behind it -- without any wait state the x-only form,
    //  which stores nothing in between, fed stale registers to the next layer: round 6, test_ode_narrow_tile_sweeps -- and the
    //  compiler does not look into the assembly.  A layer's registers are rectified back to back and fence() keeps the next layer's
    //  matrix instructions behind all of them, so only the LAST register of a layer needs the two slots spelled out: the others
    //  have the next register's two instructions behind them.  A lone wave pays ~5 clocks per s_nop: 14 per time step now, 42 before.)
    if constexpr (XW_MASK_WORDS(M) == 2) {      // (the carry out of the low word shifts into the high one)
      if (kb_last) asm("v_addc_co_u32_e64 %1, vcc, %1, %1, %4\n\tv_addc_co_u32_e64 %2, vcc, %2, %2, vcc\n\tv_max_f64 %0, %3, 0\n\ts_nop 1" : "=v"(r), "+v"(bits), "+v"(bits_hi) : "v"(z), "s"(open) : "vcc");
      else asm("v_addc_co_u32_e64 %1, vcc, %1, %1, %4\n\tv_addc_co_u32_e64 %2, vcc, %2, %2, vcc\n\tv_max_f64 %0, %3, 0" : "=v"(r), "+v"(bits), "+v"(bits_hi) : "v"(z), "s"(open) : "vcc");
      return r;
    }
    if (kb_last) asm("v_addc_co_u32_e64 %1, vcc, %1, %1, %3\n\tv_max_f64 %0, %2, 0\n\ts_nop 1" : "=v"(r), "+v"(bits) : "v"(z), "s"(open) : "vcc");
    else asm("v_addc_co_u32_e64 %1, vcc, %1, %1, %3\n\tv_max_f64 %0, %2, 0" : "=v"(r), "+v"(bits) : "v"(z), "s"(open) : "vcc");
    return r;
  }
  __device__ __forceinline__ void fence() const { __builtin_amdgcn_sched_barrier(0); }
  __device__ __forceinline__ void z(int j, d4 r) const {
    if (FULL) act_store(A, row0 + j * K, K, N, q, r);
  }
  __device__ __forceinline__ void a(d4 v) const { act_store(A, row0 + (M - 1) * K, K, N, q, v); }
};

// ACT: 0 = no activation store, 1 = the full store, 2 = only what an x-only sweep reads (tanh rows + ReLU mask words)
template <int H, int K, int M, int METHOD, int ACT>
__global__ void __launch_bounds__(64, XW_ODE_FWD_WAVES) k_ode_fwd(const FwdJobs jobs, const double* __restrict__ tf,
                                                const double* __restrict__ th, int L, int d) {
  typedef Dim<H, K> D;
  typedef RK<METHOD> T;
  const int job = find_job(jobs, (int)blockIdx.x);
  const double* __restrict__ xT = jobs.xT[job];
  const double* __restrict__ start = jobs.start[job];
  double* __restrict__ u = jobs.u[job];
  double* __restrict__ Y = jobs.Y[job];
  double* __restrict__ act = jobs.act[job];
  xw_setprio(jobs.prio);
  typedef ActLayout<H, K, M, T::S> AL;
  const int N = jobs.N[job];
  const int lane = xw_lane(), g = lane >> 4, n = lane & 15;
  const int base = ((int)blockIdx.x - jobs.tile0[job]) * 16;
  if (blockIdx.x == 0 && jobs.zero16 != nullptr && lane < 16) jobs.zero16[lane] = 0.0;
  const bool valid = base + n < N;
  const int ncl = valid ? base + n : N - 1;
  const UOff o = u_offsets(d, H, K);
  FieldW<H, K> w;
  load_field<H, K>(th, o, d, w);
  d4 a0[D::HT], a1[D::HT], y[D::HT], flw[D::HT];
  lift<H, K>(th, o, start[ncl], a0, a1, y);
  const d4 xp = project_x<H, K>(th, o, xT, N, d, ncl);
#pragma unroll
  for (int ht = 0; ht < D::HT; ++ht) flw[ht] = xw_vecD(th + o.FLw, H, 16 * ht);
  const double flb = th[o.FLb];
  const ActLane aq = act_lane(16, n, true, K);           // every lane of the tile has a slot (padding paths: finite copies)
  const long ntile = (N + 15) >> 4, tile = base >> 4;
  // The checkpoint y_l and u_l leave through buffer descriptors as well (one per time index: rows of [H][N] resp. one row of
  // u): lanes that own nothing -- paths past N in the last tile, rows >= H of the last row tile, the lane groups g > 0 of u --
  // carry an out-of-range offset, so the time loop has no branch but its own (see ActLane.off_part_st).
  const bool big = (long)H * N * 8 >= (1L << 31);         // (a descriptor's 32-bit range: beyond it the plain guarded stores)
  unsigned yoff[4 * D::HT];
#pragma unroll
  for (int ht = 0; ht < D::HT; ++ht)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * ht + g + 4 * r;
      yoff[4 * ht + r] = (valid && row < H) ? 8u * (unsigned)(g * N + base + n) : XW_ACT_OOB;
    }
  const unsigned uoff = (g == 0 && valid) ? 8u * (unsigned)(base + n) : XW_ACT_OOB;
  for (int l = 0; l < L; ++l) {
    double part = 0.0;
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(Y != nullptr && !big ? Y + (long)l * H * N : nullptr, 0,
                                                                         Y != nullptr && !big ? H * N * 8 : 0, 0x00020000);
#pragma unroll
    for (int ht = 0; ht < D::HT; ++ht)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        part += flw[ht][r] * y[ht][r];
        if (16 * ht + 4 * r < H) {                         // (compile time: the register holds live rows)
          const double x = y[ht][r];
          const xw_u2v w2 = {(unsigned)__double2loint(x), (unsigned)__double2hiint(x)};
          __builtin_amdgcn_raw_buffer_store_b64(w2, yrs, (int)yoff[4 * ht + r], (16 * ht + 4 * r) * N * 8, 0);
        }
      }
    if (big && Y != nullptr) {                             // (uniform, never taken below 13 million paths)
#pragma unroll
      for (int ht = 0; ht < D::HT; ++ht)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * ht + g + 4 * r;
          if (valid && row < H) Y[((long)l * H + row) * N + base + n] = y[ht][r];
        }
    }
    const double ul = xw_sum_over_g(part) + flb;           // final_linear, src/model.py:110
    {
      const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(u + (long)l * N, 0, N * 8, 0x00020000);
      const xw_u2v w2 = {(unsigned)__double2loint(ul), (unsigned)__double2hiint(ul)};
      __builtin_amdgcn_raw_buffer_store_b64(w2, urs, (int)uoff, 0, 0);
    }
    if (l == L - 1) break;
    const double t0 = tf[l], dt = tf[l + 1] - t0;
    d4 k[T::S][D::HT];
#pragma unroll
    for (int i = 0; i < T::S; ++i) {
      d4 yi[D::HT];
#pragma unroll
      for (int ht = 0; ht < D::HT; ++ht) {
        yi[ht] = y[ht];
#pragma unroll
        for (int j = 0; j < i; ++j)
          if (T::a(i, j) != 0.0) yi[ht] += (dt * T::a(i, j)) * k[j][ht];
      }
      if (!ACT) {
        field_fwd<H, K, M, true>(w, t0 + T::c(i) * dt, xp, yi, k[i], SinkNone{});
      } else {
        double* __restrict__ A = act + ((long)l * ntile + tile) * (AL::TOTAL * 16);
        const ActBuf AB = act_buf(A, AL::TOTAL * 16 * 8);
        if (i > 0 && ACT == 1) {
#pragma unroll
          for (int ht = 0; ht < D::HT; ++ht)
            act_store(AB, AL::YI + (i - 1) * H + 16 * ht, H - 16 * ht < 16 ? H - 16 * ht : 16, 16, aq, yi[ht]);
        }
        unsigned bits = 0, bits_hi = 0;
        field_fwd<H, K, M, true>(w, t0 + T::c(i) * dt, xp, yi, k[i], SinkAct<K, M, ACT == 1>{AB, i * AL::STAGE, 16, aq, bits, bits_hi});
        __builtin_amdgcn_raw_buffer_store_b32(bits, AB.rsrc, lane * 4, (AL::MASK + AL::MR * i) * 16 * 8, 0);   // 1 = open (SaveX)
        if (XW_MASK_WORDS(M) == 2) __builtin_amdgcn_raw_buffer_store_b32(bits_hi, AB.rsrc, lane * 4, (AL::MASK + AL::MR * i + 2) * 16 * 8, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < T::S; ++i)
      if (T::b(i) != 0.0)
#pragma unroll
        for (int ht = 0; ht < D::HT; ++ht) y[ht] += (dt * T::b(i)) * k[i][ht];
  }
}

// store a chain-layout accumulator tile (rows r0.., cols c0..) into a row-major matrix
__device__ __forceinline__ void storeD(double* dst, int ld, int rows, int cols, int r0, int c0, d4 acc) {
  const int lane = xw_lane(), g = lane >> 4, c = c0 + (lane & 15);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = r0 + g + 4 * r;
    if (row < rows && c < cols) dst[row * ld + c] = acc[r];
  }
}
// store column `col` of a chain-layout accumulator tile as a (strided) vector
__device__ __forceinline__ void storeDcol(double* dst, int stride, int rows, int r0, int col, d4 acc) {
  const int lane = xw_lane(), g = lane >> 4;
  if ((lane & 15) != col) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = r0 + g + 4 * r;
    if (row < rows) dst[(long)row * stride] = acc[r];
  }
}
// row sums over the 16 paths of a chain tile -> vector
__device__ __forceinline__ void storeRowSums(double* dst, int rows, int r0, d4 q) {
  const int lane = xw_lane(), g = lane >> 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const double s = xw_sum_over_n(q[r]);
    const int row = r0 + g + 4 * r;
    if ((lane & 15) == 0 && row < rows) dst[row] = s;
  }
}

#ifndef XW_ODE_WIDE16
// the field's weight-gradient accumulators -> one slab
template <int H, int K, bool HID = true, bool IO = true>
__device__ __forceinline__ void store_field_grads(double* slab, const UOff& o, int d, const FieldG<H, K>& G) {
  typedef Dim<H, K> D;
  if (HID) {
    storeD(slab + o.Wh, K, K, K, 0, 0, G.Wh);
    storeDcol(slab + o.Whb, 1, K, 0, K, G.Wh);
  }
  if (!IO) return;
#pragma unroll
  for (int ct = 0; ct < (H + 1 + 15) / 16; ++ct) {
    storeD(slab + o.Win + d + 1, o.ldin, K, H, 0, 16 * ct, G.Wy[ct]);
    if (ct == (H >> 4)) storeDcol(slab + o.Win + d, o.ldin, K, 0, H & 15, G.Wy[ct]);
  }
#pragma unroll
  for (int ht = 0; ht < D::HT; ++ht) {
    storeD(slab + o.Wo, K, H, K, 16 * ht, 0, G.Wo[ht]);
    storeDcol(slab + o.Wob, 1, H, 16 * ht, K, G.Wo[ht]);
  }
}

#else
// the field's weight-gradient accumulators -> one slab (wide container: the biases and the time column are row sums over the 16 paths)
template <int H, int K, bool HID = true, bool IO = true>
__device__ __forceinline__ void store_field_grads(double* slab, const UOff& o, int d, const FieldG<H, K>& G) {
  typedef Dim<H, K> D;
  const int lane = xw_lane(), g = lane >> 4;
  if (HID) {
    storeD(slab + o.Wh, K, K, K, 0, 0, G.Wh);
    storeRowSums(slab + o.Whb, K, 0, G.bh);
  }
  if (!IO) return;
#pragma unroll
  for (int ct = 0; ct < D::HT; ++ct) storeD(slab + o.Win + d + 1, o.ldin, K, H, 0, 16 * ct, G.Wy[ct]);
#pragma unroll
  for (int r = 0; r < 4; ++r) {                               // Win[:, d]: the time column
    const double s_ = xw_sum_over_n(G.wt[r]);
    const int row = g + 4 * r;
    if ((lane & 15) == 0 && row < K) slab[o.Win + (long)row * o.ldin + d] = s_;
  }
#pragma unroll
  for (int ht = 0; ht < D::HT; ++ht) {
    storeD(slab + o.Wo, K, H, K, 16 * ht, 0, G.Wo[ht]);
    storeRowSums(slab + o.Wob, H, 16 * ht, G.bo[ht]);
  }
}

#endif
// what the reverse of one step l -> l+1 needs from the forward pass: the stage inputs and the stage activations
template <int H, int K, int M, int S> struct Rec {
  d4 yi[S][Dim<H, K>::HT];
  Save<M> sv[S];
};
template <int H, int K>
__device__ __forceinline__ void load_ckpt(const double* __restrict__ Y, int l, int N, int ncl, d4 (&y)[Dim<H, K>::HT]) {
  const int g = xw_lane() >> 4;
#pragma unroll
  for (int ht = 0; ht < Dim<H, K>::HT; ++ht)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * ht + g + 4 * r;
      y[ht][r] = row < H ? Y[((long)l * H + row) * N + ncl] : 0.0;
    }
}
// rebuild the stages of step l -> l+1 from the checkpoint y_l (no dependence on any cotangent)
template <int H, int K, int M, int METHOD>
__device__ __forceinline__ void recompute(const FieldW<H, K>& w, d4 xp, const double* __restrict__ Y,
                                          const double* __restrict__ tf, int l, int N, int ncl,
                                          Rec<H, K, M, RK<METHOD>::S>& R) {
  typedef Dim<H, K> D;
  typedef RK<METHOD> T;
  const double t0 = tf[l], dt = tf[l + 1] - t0;
  d4 k[T::S][D::HT];
  load_ckpt<H, K>(Y, l, N, ncl, R.yi[0]);
#pragma unroll
  for (int i = 0; i < T::S; ++i) {
    if (i > 0) {
#pragma unroll
      for (int ht = 0; ht < D::HT; ++ht) {
        R.yi[i][ht] = R.yi[0][ht];
#pragma unroll
        for (int j = 0; j < i; ++j)
          if (T::a(i, j) != 0.0) t_axpy(R.yi[i][ht], dt * T::a(i, j), k[j][ht], D::HR(ht));
      }
    }
    if (i < T::S - 1)
      field_fwd<H, K, M, true>(w, t0 + T::c(i) * dt, xp, R.yi[i], k[i], SinkSave<M>{R.sv[i]});
    else
      field_fwd<H, K, M, false>(w, t0 + T::c(i) * dt, xp, R.yi[i], k[i], SinkSave<M>{R.sv[i]});   // last stage: activations only
  }
}

// one stage of that record read back from the forward pass's activation store (k_ode_fwd<ACT>): loads only.
// Stage granularity: while one stage is reversed the loads of the next one are in flight -- a whole step ahead would
// keep twice as many registers occupied.
template <int H, int K, int M, class SV = Save<M>> struct StageRec {
  d4 yi[Dim<H, K>::HT];        // (not loaded for SaveX: a sweep without weight gradients never reads the stage input)
  SV sv;
};
template <int H, int K, int M, int METHOD, class SV>
__device__ __forceinline__ void load_stage(const double* __restrict__ Y, const double* __restrict__ act, int l, int i,
                                           int N, int ncl, StageRec<H, K, M, SV>& R) {
  typedef Dim<H, K> D;
  typedef ActLayout<H, K, M, RK<METHOD>::S> AL;
  const long ntile = (N + 15) >> 4, tile = __builtin_amdgcn_readfirstlane(ncl >> 4);   // (clamped lanes stay in the tile)
  const double* __restrict__ A = act + ((long)l * ntile + tile) * (AL::TOTAL * 16);
  const ActLane q = act_lane(16, xw_lane() & 15, true, K);
  if constexpr (std::is_same<SV, SaveX<M>>::value) {
    R.sv.bits = reinterpret_cast<const unsigned*>(A + (AL::MASK + AL::MR * i) * 16)[xw_lane()];
    R.sv.bits_hi = XW_MASK_WORDS(M) == 2 ? reinterpret_cast<const unsigned*>(A + (AL::MASK + AL::MR * i + 2) * 16)[xw_lane()] : 0u;
    R.sv.a = act_load(A, i * AL::STAGE + (M - 1) * K, K, 16, q);
    return;
  }
  if (i == 0) {
    load_ckpt<H, K>(Y, l, N, ncl, R.yi);
  } else {
#pragma unroll
    for (int ht = 0; ht < D::HT; ++ht)
      R.yi[ht] = act_load(A, AL::YI + (i - 1) * H + 16 * ht, H - 16 * ht < 16 ? H - 16 * ht : 16, 16, q);
  }
  if constexpr (std::is_same<SV, Save<M>>::value) {
#pragma unroll
    for (int j = 0; j < M - 1; ++j) R.sv.z[j] = act_load(A, i * AL::STAGE + j * K, K, 16, q);
    R.sv.a = act_load(A, i * AL::STAGE + (M - 1) * K, K, 16, q);
  }
}

// ---- end of a sweep: everything behind the time loop, for ONE tile of 16 paths in the chain layout ------------------------
// x-projection (cotangent of x, gradients of Win[:, :d] and Win.b), read-out layer gradients, the lift (initial layers):
// lam = cotangent of y_0, xpb = sum of the cotangents of z_0 over all field evaluations, ub0 = cotangent of u at the first
// time index, accFL / accFLb = sums of ubar y_l / ubar.  Shared by the 16-path sweeps (sweep_body) and by the narrow-tile
// sweep (xw_ode_n4.h), whose four waves hand these registers to their first wave through LDS.  lds: XW_SWEEP_TILES tiles.
template <int H, int K, bool PARAMS, bool ADJ, class StoreFieldGrads>
__device__ __forceinline__ void sweep_tail(const double* __restrict__ th, const UOff& o, int d, int N, int base, bool valid, int ncl,
                                           const double* __restrict__ xT, double sv, bool x_ones, const d4 (&lam)[Dim<H, K>::HT],
                                           d4 xpb, double ub0, const d4 (&accFL)[Dim<H, K>::HT], double accFLb,
                                           const d4 (&flw)[Dim<H, K>::HT], double* __restrict__ gx, double* __restrict__ gs,
                                           double* slab, double* lds, const StoreFieldGrads& store_field) {
  typedef Dim<H, K> D;
  const int lane = xw_lane(), g = lane >> 4, n = lane & 15;
  // ---- x-projection: cotangent of x, gradients of Win[:, :d] and Win.b -----------------------------------------
  if (gx != nullptr) {
    for (int rt = 0; rt < (d + 15) / 16; ++rt) {
      d4 v = xw_zero4();
      if (!ADJ) {                // (odeint_adjoint: x is not among the inputs the adjoint differentiates with respect to)
#pragma unroll
        for (int ks = 0; ks < D::KSK; ++ks) v = XW_MFMA(xw_fragAT(th + o.Win, o.ldin, K, d, 16 * rt, 4 * ks), xpb[ks], v);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * rt + g + 4 * r;
        if (i < d && valid) gx[(long)i * N + base + n] = v[r];
      }
    }
  }
  if (PARAMS) {
    xw_writeT(lds, xpb);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int ct = 0; ct < (d + 15) / 16; ++ct) {
      d4 acc = xw_zero4();
      const int i = 16 * ct + (lane & 15);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int pn = base + 4 * ks + (lane >> 4);
        const double b = (i < d && pn < N) ? xT[(long)i * N + pn] : 0.0;
        acc = XW_MFMA(xw_readT(lds, ks), b, acc);
      }
      storeD(slab + o.Win, o.ldin, K, d, 0, 16 * ct, acc);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    storeRowSums(slab + o.Winb, K, 0, xpb);
    store_field();                                         // (the field's own accumulators, where their owner keeps them)
#pragma unroll
    for (int ht = 0; ht < D::HT; ++ht) storeRowSums(slab + o.FLw, H, 16 * ht, accFL[ht]);
    const double sb = xw_sum_over_n(accFLb);
    if (lane == 0) slab[o.FLb] = sb;
  }

  // ---- initial layers: lam = cotangent of y_0 ----------------------------------------------------------------------
  {
    d4 a0[D::HT], a1[D::HT], y0[D::HT];
    lift<H, K>(th, o, sv, a0, a1, y0);
    d4 d1[D::HT], d0[D::HT];
    if (PARAMS) {
#pragma unroll
      for (int rt = 0; rt < D::HT; ++rt) {
#pragma unroll
        for (int ct = 0; ct < D::HT; ++ct) {
          d4 acc = xw_zero4();
          outer_acc(acc, lam[rt], a1[ct], lds);
          storeD(slab + o.IL4w, H, H, H, 16 * rt, 16 * ct, acc);
        }
        storeRowSums(slab + o.IL4b, H, 16 * rt, lam[rt]);
      }
    }
#pragma unroll
    for (int ht = 0; ht < D::HT; ++ht) {
      d4 v = xw_zero4();
#pragma unroll
      for (int ks = 0; ks < D::KSH; ++ks) v = XW_MFMA(xw_fragAT(th + o.IL4w, H, H, H, 16 * ht, 4 * ks), lam[ks >> 2][ks & 3], v);
#pragma unroll
      for (int r = 0; r < 4; ++r) d1[ht][r] = a1[ht][r] > 0.0 ? v[r] : 0.0;
    }
    if (PARAMS) {
#pragma unroll
      for (int rt = 0; rt < D::HT; ++rt) {
#pragma unroll
        for (int ct = 0; ct < D::HT; ++ct) {
          d4 acc = xw_zero4();
          outer_acc(acc, d1[rt], a0[ct], lds);
          storeD(slab + o.IL2w, H, H, H, 16 * rt, 16 * ct, acc);
        }
        storeRowSums(slab + o.IL2b, H, 16 * rt, d1[rt]);
      }
    }
    double gpart = 0.0;
#pragma unroll
    for (int ht = 0; ht < D::HT; ++ht) {
      d4 v = xw_zero4();
#pragma unroll
      for (int ks = 0; ks < D::KSH; ++ks) v = XW_MFMA(xw_fragAT(th + o.IL2w, H, H, H, 16 * ht, 4 * ks), d1[ks >> 2][ks & 3], v);
      const d4 w0 = xw_vecD(th + o.IL0w, H, 16 * ht);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        d0[ht][r] = a0[ht][r] > 0.0 ? v[r] : 0.0;
        gpart += w0[r] * d0[ht][r];
      }
      if (PARAMS) {
        storeRowSums(slab + o.IL0w, H, 16 * ht, d0[ht] * sv);
        storeRowSums(slab + o.IL0b, H, 16 * ht, d0[ht]);
      }
    }
    double gsn = xw_sum_over_g(gpart);
    if (PARAMS && gs != nullptr && x_ones) {
      // The x outputs of this launch stand for the helper backward u.backward(ones) (src/loss.py:55) while the
      // parameter gradients use ubar, which differs from ones at the first time index only (the initial-value penalty).
      // gx never sees that entry (no step is reversed after it), d/d start does, linearly through the lift:
      // subtract the lift's response to flw * (ubar_0 - 1).
      const double dub = ub0 - (valid ? 1.0 : 0.0);
      d4 e1[D::HT];
#pragma unroll
      for (int ht = 0; ht < D::HT; ++ht) {
        d4 v = xw_zero4();
#pragma unroll
        for (int ks = 0; ks < D::KSH; ++ks) v = XW_MFMA(xw_fragAT(th + o.IL4w, H, H, H, 16 * ht, 4 * ks), flw[ks >> 2][ks & 3], v);
#pragma unroll
        for (int r = 0; r < 4; ++r) e1[ht][r] = a1[ht][r] > 0.0 ? v[r] : 0.0;
      }
      double cpart = 0.0;
#pragma unroll
      for (int ht = 0; ht < D::HT; ++ht) {
        d4 v = xw_zero4();
#pragma unroll
        for (int ks = 0; ks < D::KSH; ++ks) v = XW_MFMA(xw_fragAT(th + o.IL2w, H, H, H, 16 * ht, 4 * ks), e1[ks >> 2][ks & 3], v);
        const d4 w0 = xw_vecD(th + o.IL0w, H, 16 * ht);
#pragma unroll
        for (int r = 0; r < 4; ++r) cpart += a0[ht][r] > 0.0 ? w0[r] * v[r] : 0.0;
      }
      gsn -= dub * xw_sum_over_g(cpart);
    }
    if (gs != nullptr && g == 0 && valid) gs[base + n] = gsn;
  }
}

// SAVED: the stage activations come from the forward pass's store (euler, midpoint); otherwise they are recomputed
// ADJ: the reference's adjoint=True (src/model.py:103: torchdiffeq.odeint_adjoint) -- not the reverse of the steps that
//   were taken but the continuous adjoint, integrated with the same fixed-grid method.  torchdiffeq 0.1.1 (absent here,
//   restated from its published OdeintAdjointMethod.backward): for i = L-1 .. 1 the augmented state (y, a, theta-bar) starts
//   from the FORWARD solution y(t_i) and takes ONE step of `method` from t_i to t_{i-1} (h < 0) under
//       d/dt (y, a, theta-bar) = ( f(t, y), -a^T df/dy, -a^T df/dtheta ),
//   then a += the cotangent of y(t_{i-1}).  Its parameter set is the field module's parameters; the sample point x is a
//   plain attribute of that module, so no gradient reaches x through the field (only through the start value).
// DUO (with PARAMS and SAVED): the "duo" sweep.  This wave runs the adjoint chain exactly like the sweep without weight
//   gradients (ReLU mask words + tanh rows from the store) and posts the cotangent tiles of every field evaluation into
//   one of two alternating LDS buffers (qbuf); a second wave of the block (duo_outer) contracts them with the layer
//   inputs, which it loads from the activation store directly in operand layout -- one workgroup barrier per field
//   evaluation.  One wave per 16 paths left three quarters of the SIMDs idle at N = 4096 and bound the sweep by the
//   latency of ONE instruction stream (chain, LDS round trips, loads); the split doubles the waves and halves the stream.
template <int H, int K, int M, int METHOD, bool PARAMS, bool SAVED, bool ADJ, bool DUO>
__device__ __forceinline__ void sweep_body(const BwdJobs& jobs, const double* __restrict__ tf, const double* __restrict__ th,
                                           int L, int d, double* lds, double* qbuf, int vb) {
  static_assert(!SAVED || RK<METHOD>::S <= 2, "the activation store is used by euler and midpoint");
  static_assert(!(SAVED && ADJ), "the continuous adjoint evaluates the field at its own stage points");
  static_assert(!DUO || (PARAMS && SAVED), "the duo sweep is the sweep with weight gradients from the activation store");
  typedef Dim<H, K> D;
  typedef RK<METHOD> T;
  constexpr int OUTER = DUO ? 2 : (PARAMS ? 1 : 0);
  xw_setprio(jobs.prio);
  int qflip = 0;                                 // DUO: which of the two Q buffers the next field evaluation posts into
  if (PARAMS && !DUO) {
    if (xw_lane() < 16) lds[2 * XW_TTILE + K * XW_TSTRIDE + xw_lane()] = 1.0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  const int job = find_job(jobs, vb);
  const double* __restrict__ xT = jobs.xT[job];
  const double* __restrict__ start = jobs.start[job];
  const double* __restrict__ Y = jobs.Y[job];
  const double* __restrict__ act = jobs.act[job];
  const double* __restrict__ ubar = jobs.ubar[job];
  double* __restrict__ gx = jobs.gx[job];
  double* __restrict__ gs = jobs.gs[job];
  double* __restrict__ gslab = jobs.gslab[job];
  const int N = jobs.N[job];
  const int tile = vb - jobs.tile0[job];
  const int lane = xw_lane(), g = lane >> 4, n = lane & 15;
  const int base = tile * 16;
  const bool valid = base + n < N;
  const int ncl = valid ? base + n : N - 1;
  const UOff o = u_offsets(d, H, K);
  FieldW<H, K> w;                               // (only the recomputing paths load it)
  FieldWT<H, K> wT;
  load_field_T<H, K>(th, o, d, wT);
  d4 flw[D::HT];
#pragma unroll
  for (int ht = 0; ht < D::HT; ++ht) flw[ht] = xw_vecD(th + o.FLw, H, 16 * ht);

  FieldG<H, K> G;
  G.Wh = xw_zero4();
#pragma unroll
  for (int ct = 0; ct < (H + 1 + 15) / 16; ++ct) G.Wy[ct] = xw_zero4();
#pragma unroll
  for (int ht = 0; ht < D::HT; ++ht) G.Wo[ht] = xw_zero4();
#ifdef XW_ODE_WIDE16
  G.bh = G.wt = xw_zero4();
#pragma unroll
  for (int ht = 0; ht < D::HT; ++ht) G.bo[ht] = xw_zero4();
#endif
  d4 accFL[D::HT];
#pragma unroll
  for (int ht = 0; ht < D::HT; ++ht) accFL[ht] = xw_zero4();
  double accFLb = 0.0;
  d4 lam[D::HT];
#pragma unroll
  for (int ht = 0; ht < D::HT; ++ht) lam[ht] = xw_zero4();
  d4 xpb = xw_zero4();
  double ub0 = 0.0;            // cotangent of u at the first time index (the last one visited)

  // read-out u_l = FL y_l + b at time index l: cotangent into lam, gradients of the read-out layer
  // cotangent of u at time index l.  Loaded at the START of the step that ends with its read-out: behind the fences of
  // the outer products the load could not be hoisted and its HBM latency sat on the chain once per step.
  auto load_ub = [&](int l) -> double {
    return valid ? cot_u(jobs, job, ubar, l, N, base + n, L) : 0.0;
  };
  auto readout = [&](int l, const d4 (&yl)[D::HT], double ub) {
    ub0 = ub;
#pragma unroll
    for (int ht = 0; ht < D::HT; ++ht) {
      t_axpy(lam[ht], ub, flw[ht], D::HR(ht));
      if (PARAMS) t_axpy(accFL[ht], ub, yl[ht], D::HR(ht));
    }
    if (PARAMS) accFLb += ub;
  };
  // reverse of ONE stage i of the step l -> l+1: kb[i] is the cotangent of the stage derivative k_i
  d4 kb[T::S][D::HT], psum[D::HT];
  auto begin_step = [&](int l) {
    const double dt = tf[l + 1] - tf[l];
#pragma unroll
    for (int i = 0; i < T::S; ++i)
#pragma unroll
      for (int ht = 0; ht < D::HT; ++ht) {
        t_scale(kb[i][ht], dt * T::b(i), lam[ht], D::HR(ht));
        if (i == 0) psum[ht] = xw_zero4();
      }
  };
  auto reverse_stage = [&](int l, int i, const d4 (&yin)[D::HT], const auto& sv) {
    const double t0 = tf[l], dt = tf[l + 1] - t0;
    d4 psi[D::HT];
    field_vjp<H, K, M, OUTER>(w, wT, t0 + T::c(i) * dt, sv, yin, kb[i], psi, xpb, G,
                              DUO ? qbuf + qflip * DuoPlan<H, K, M>::BUF : lds);
    if (DUO) {      // the tiles of this evaluation are complete: hand them over (the partner is one evaluation behind)
      // (LDS only: a workgroup-scope release fence would also drain vmcnt, i.e. wait for the next stage's prefetch loads)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      qflip ^= 1;
    }
#pragma unroll
    for (int ht = 0; ht < D::HT; ++ht) {
      t_add(psum[ht], psi[ht], D::HR(ht));
#pragma unroll
      for (int j = 0; j < T::S; ++j)
        if (j < i && T::a(i, j) != 0.0) t_axpy(kb[j][ht], dt * T::a(i, j), psi[ht], D::HR(ht));
    }
  };
  auto end_step = [&](int l, const d4 (&y_l)[D::HT], double ub) {
#pragma unroll
    for (int ht = 0; ht < D::HT; ++ht) t_add(lam[ht], psum[ht], D::HR(ht));
    readout(l, y_l, ub);
  };

  if constexpr (ADJ) {
    load_field<H, K>(th, o, d, w);
    const d4 xp = project_x<H, K>(th, o, xT, N, d, ncl);
    {
      d4 yl[D::HT];
      load_ckpt<H, K>(Y, L - 1, N, ncl, yl);
      readout(L - 1, yl, load_ub(L - 1));
    }
    for (int l = L - 2; l >= 0; --l) {
      const double ub = load_ub(l);
      const double t1 = tf[l + 1], dt = t1 - tf[l];             // the step goes from t1 back to t1 - dt
      d4 y1[D::HT];
      load_ckpt<H, K>(Y, l + 1, N, ncl, y1);
      d4 ky[T::S][D::HT], ps[T::S][D::HT], asum[D::HT];          // stage derivatives of y; a_s^T df/dy; update of the adjoint
#pragma unroll
      for (int ht = 0; ht < D::HT; ++ht) asum[ht] = xw_zero4();
#pragma unroll
      for (int i = 0; i < T::S; ++i) {
        d4 ys[D::HT], as[D::HT];
#pragma unroll
        for (int ht = 0; ht < D::HT; ++ht) {
          ys[ht] = y1[ht];
          as[ht] = lam[ht];
#pragma unroll
          for (int j = 0; j < i; ++j)
            if (T::a(i, j) != 0.0) {
              ys[ht] -= (dt * T::a(i, j)) * ky[j][ht];           // h = -dt;  dy/dt = f
              as[ht] += (dt * T::a(i, j)) * ps[j][ht];           //           da/dt = -a^T df/dy
            }
        }
        const double ti = t1 - T::c(i) * dt;
        Save<M> sv;
        field_fwd<H, K, M, true>(w, ti, xp, ys, ky[i], SinkSave<M>{sv});
        if (T::b(i) != 0.0) {
          // this stage enters the update: the vector-Jacobian product is taken of (dt b_i) a_s, so that the weight-gradient
          // outer products inside it accumulate  -h b_i a_s^T df/dtheta  directly
          d4 cot[D::HT], psi[D::HT];
#pragma unroll
          for (int ht = 0; ht < D::HT; ++ht) cot[ht] = (dt * T::b(i)) * as[ht];
          field_vjp<H, K, M, PARAMS ? 1 : 0>(w, wT, ti, sv, ys, cot, psi, xpb, G, lds);
#pragma unroll
          for (int ht = 0; ht < D::HT; ++ht) {
            asum[ht] += psi[ht];
            ps[i][ht] = (1.0 / (dt * T::b(i))) * psi[ht];
          }
        } else {
          d4 xdummy = xw_zero4();
          FieldG<H, K> Gd;                                       // (untouched: no parameter products in this call)
          field_vjp<H, K, M, 0>(w, wT, ti, sv, ys, as, ps[i], xdummy, Gd, lds);
        }
      }
#pragma unroll
      for (int ht = 0; ht < D::HT; ++ht) lam[ht] += asum[ht];
      d4 yl[D::HT];
      load_ckpt<H, K>(Y, l, N, ncl, yl);
      readout(l, yl, ub);
    }
  } else if constexpr (SAVED) {
    // (without weight gradients the sweep only needs tanh(z_{m-1}) and the ReLU masks of a stage)
    // (the duo sweep's chain wave neither: its partner reads the layer inputs)
    typedef typename std::conditional<PARAMS && !DUO, Save<M>, SaveX<M>>::type SV;
    // y_l for the read-out layer's weight gradient: the stage-0 record carries it, except in the duo sweep
    auto y_of = [&](int l, const d4 (&rec)[D::HT], d4 (&yl)[D::HT]) {
      if (DUO) load_ckpt<H, K>(Y, l, N, ncl, yl);
      else {
#pragma unroll
        for (int ht = 0; ht < D::HT; ++ht) yl[ht] = rec[ht];
      }
    };
    // The cotangent of u is requested ONE STEP AHEAD by straight-line code (Cot4 above), in front of the stage record's
    // requests: cot_u's branches wait for their loads where they stand (vmcnt(0)), which drained the record prefetched for the
    // next stage -- one exposed memory latency per step in every sweep whose cotangent is formed from a residual (the
    // boundary sweep, sweep B).
    const Cot4 cot = make_cot(jobs, job, N, ncl, valid, Y);
    auto run = [&](auto weak_tag) {
      constexpr bool WEAK = decltype(weak_tag)::value;
      StageRec<H, K, M, SV> sa, sb;
      const CotRaw raw_last = cot_issue<WEAK>(cot, L - 1);
      CotRaw raw = cot_issue<WEAK>(cot, L > 1 ? L - 2 : 0);
      if constexpr (T::S == 2) {
        // stage 1 always lives in sa, stage 0 in sb: each is loaded while the other one is reversed
        if (L > 1) load_stage<H, K, M, METHOD>(Y, act, L - 2, 1, N, ncl, sa);
      } else {
        if (L > 1) load_stage<H, K, M, METHOD>(Y, act, L - 2, 0, N, ncl, sa);
      }
      {
        d4 yl[D::HT];
        load_ckpt<H, K>(Y, L - 1, N, ncl, yl);
        readout(L - 1, yl, cot_value<WEAK>(cot, raw_last, L - 1, L));
      }
      if constexpr (T::S == 2) {
        for (int l = L - 2; l >= 0; --l) {
          const double ub = cot_value<WEAK>(cot, raw, l, L);       // (requested a step ago)
          d4 yl[D::HT];
          load_stage<H, K, M, METHOD>(Y, act, l, 0, N, ncl, sb);
          raw = cot_issue<WEAK>(cot, l > 0 ? l - 1 : 0);
          y_of(l, sb.yi, yl);
          begin_step(l);
          reverse_stage(l, 1, sa.yi, sa.sv);
          load_stage<H, K, M, METHOD>(Y, act, l > 0 ? l - 1 : 0, 1, N, ncl, sa);
          reverse_stage(l, 0, sb.yi, sb.sv);
          end_step(l, yl, ub);                              // stage 0's input is y_l itself
        }
      } else {
        for (int l = L - 2; l >= 0; --l) {
          const double ub = cot_value<WEAK>(cot, raw, l, L);
          d4 yl[D::HT];
          sb = sa;
          raw = cot_issue<WEAK>(cot, l > 0 ? l - 1 : 0);
          load_stage<H, K, M, METHOD>(Y, act, l > 0 ? l - 1 : 0, 0, N, ncl, sa);
          y_of(l, sb.yi, yl);
          begin_step(l); reverse_stage(l, 0, sb.yi, sb.sv); end_step(l, yl, ub);
        }
      }
    };
    if (cot.weak) run(std::true_type{});
    else run(std::false_type{});
  } else if constexpr (T::S <= 2) {
    load_field<H, K>(th, o, d, w);
    const d4 xp = project_x<H, K>(th, o, xT, N, d, ncl);
    Rec<H, K, M, T::S> cur;
    {
      d4 yl[D::HT];
      load_ckpt<H, K>(Y, L - 1, N, ncl, yl);
      readout(L - 1, yl, load_ub(L - 1));
    }
    for (int l = L - 2; l >= 0; --l) {
      const double ub = load_ub(l);
      recompute<H, K, M, METHOD>(w, xp, Y, tf, l, N, ncl, cur);
      begin_step(l);
#pragma unroll
      for (int i = T::S - 1; i >= 0; --i) reverse_stage(l, i, cur.yi[i], cur.sv[i]);
      end_step(l, cur.yi[0], ub);
    }
  } else {
    load_field<H, K>(th, o, d, w);
    const d4 xp = project_x<H, K>(th, o, xT, N, d, ncl);
    for (int l = L - 1; l >= 0; --l) {
      d4 y[D::HT];
  #pragma unroll
      for (int ht = 0; ht < D::HT; ++ht)
  #pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * ht + g + 4 * r;
          y[ht][r] = row < H ? Y[((long)l * H + row) * N + ncl] : 0.0;
        }
      if (l < L - 1) {
        // reverse of the step l -> l+1 : lam currently holds the total cotangent of y_{l+1}
        const double t0 = tf[l], dt = tf[l + 1] - t0;
        d4 k[T::S][D::HT], kb[T::S][D::HT], psum[D::HT];
        Save<M> sv;
  #pragma unroll
        for (int i = 0; i < T::S - 1; ++i) {      // stage derivatives needed to rebuild the later stage inputs
          d4 yi[D::HT];
  #pragma unroll
          for (int ht = 0; ht < D::HT; ++ht) {
            yi[ht] = y[ht];
  #pragma unroll
            for (int j = 0; j < i; ++j)
              if (T::a(i, j) != 0.0) yi[ht] += (dt * T::a(i, j)) * k[j][ht];
          }
          field_fwd<H, K, M, true>(w, t0 + T::c(i) * dt, xp, yi, k[i], SinkNone{});
        }
  #pragma unroll
        for (int i = 0; i < T::S; ++i)
  #pragma unroll
          for (int ht = 0; ht < D::HT; ++ht) {
            kb[i][ht] = (dt * T::b(i)) * lam[ht];
            if (i == 0) psum[ht] = xw_zero4();
          }
  #pragma unroll
        for (int i = T::S - 1; i >= 0; --i) {
          d4 yi[D::HT], ko[D::HT], psi[D::HT];
  #pragma unroll
          for (int ht = 0; ht < D::HT; ++ht) {
            yi[ht] = y[ht];
  #pragma unroll
            for (int j = 0; j < i; ++j)
              if (T::a(i, j) != 0.0) yi[ht] += (dt * T::a(i, j)) * k[j][ht];
          }
          const double ti = t0 + T::c(i) * dt;
          field_fwd<H, K, M, true>(w, ti, xp, yi, ko, SinkSave<M>{sv});
          field_vjp<H, K, M, PARAMS ? 1 : 0>(w, wT, ti, sv, yi, kb[i], psi, xpb, G, lds);
  #pragma unroll
          for (int ht = 0; ht < D::HT; ++ht) {
            psum[ht] += psi[ht];
  #pragma unroll
            for (int j = 0; j < i; ++j)
              if (T::a(i, j) != 0.0) kb[j][ht] += (dt * T::a(i, j)) * psi[ht];
          }
        }
  #pragma unroll
        for (int ht = 0; ht < D::HT; ++ht) lam[ht] += psum[ht];
      }
      // read-out u_l = FL y_l + b
      const double ub = valid ? cot_u(jobs, job, ubar, l, N, base + n, L) : 0.0;
      ub0 = ub;
  #pragma unroll
      for (int ht = 0; ht < D::HT; ++ht) {
        lam[ht] += flw[ht] * ub;
        if (PARAMS) accFL[ht] += y[ht] * ub;
      }
      if (PARAMS) accFLb += ub;
    }
  }

  // (duo sweep: the three transpose tiles of the epilogue live in the Q buffer that is NOT in flight -- the partners are
  //  reading the one the last evaluation was posted into; they finished with the other one before the last barrier)
  if (DUO) lds = qbuf + qflip * DuoPlan<H, K, M>::BUF;
  double* slab = PARAMS ? gslab + (long)tile * o.total : nullptr;
  sweep_tail<H, K, PARAMS, ADJ>(th, o, d, N, base, valid, ncl, xT, start[ncl], jobs.x_ones != 0, lam, xpb, ub0, accFL, accFLb, flw,
                                gx, gs, slab, lds, [&]() {
                                  if (PARAMS && !DUO) store_field_grads<H, K>(slab, o, d, G);   // (duo sweep: the partner wave stores them)
                                });
}

#ifndef XW_ODE_WIDE16
// ---- the duo sweep's second wave: weight gradients of the field -------------------------------------------------------
// For every field evaluation (same order as the chain wave, one evaluation behind it):
//     dWo += cot(out) (x) [tanh(z_{m-1}) ; 1]     dWh += sum_j cot(z_{j+1}) (x) [relu(z_j) ; 1]     dWy += cot(z_0) (x) [y_in ; t]
// contractions over the 16 paths of the tile, on v_mfma_f64_4x4x4_4b_f64 with the instruction's four blocks = the four
// GROUPS OF FOUR PATHS:  acc[g][i][j] += sum_{k<4} q[4 rb + i][path 4 g + k] * r[4 cb + j][path 4 g + k].  One instruction
// covers a 4 x 4 block of the gradient over ALL 16 paths (every block does useful work, 4-row / 4-column granularity: dWh
// is 3 x 3 instructions per layer, 76 % of their multiply-adds useful, against 4 16x16x4 instructions at 43 %: 96 x 18
// instead of 44 x 66 clocks per evaluation); the four per-group partial sums of an accumulator are added ONCE, at the end
// of the sweep.  A operands: the cotangent tiles the chain wave posted (transposed) in LDS, lane (i, g, k) reads row
// 4 rb + i, path 4 g + k.  B operands: the layer inputs straight from the activation store / the checkpoints in that same
// lane layout, fetched a whole evaluation ahead (a register is reloaded for the next evaluation right behind the last
// instruction that reads it).  Rows a block has no data for (the ones row that collects the bias gradient, zero padding)
// read a constant table -- every lane loads, the loop stays ONE basic block (see duo_b_ptr).
typedef const double __attribute__((address_space(1)))* xw_gptr;
__device__ const double xw_duo_const[2][16] = {{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
                                               {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1}};
// where the operands of field evaluation e (chain-wave order: steps L-2 .. 0, stages S-1 .. 0) come from
template <int H, int K, int M, int METHOD> struct DuoSrc {
  const double* __restrict__ A;      // activation record of the step
  const double* __restrict__ Yl;     // checkpoint y_l of the step (stage 0's input)
  int i;                             // stage
  double ti;                         // its time
  __device__ __forceinline__ DuoSrc(const double* __restrict__ Y, const double* __restrict__ act,
                                    const double* __restrict__ tf, int e, int L, int N, int tile) {
    typedef RK<METHOD> T;
    typedef ActLayout<H, K, M, T::S> AL;
    const int l = L - 2 - e / T::S;
    i = T::S - 1 - e % T::S;
    A = act + ((long)l * ((N + 15) >> 4) + tile) * (AL::TOTAL * 16);
    Yl = Y + (long)l * H * N;
    const double t0 = tf[l];
    ti = t0 + T::c(i) * (tf[l + 1] - t0);
  }
};
template <int H, int K, int M> struct Duo4 {
  static constexpr int KB1 = (K + 1 + 3) / 4;     // column blocks of [layer input ; 1]
  static constexpr int HB1 = (H + 1 + 3) / 4;     // column blocks of [y ; t]
  static constexpr int KB = Dim<H, K>::KB, HB = Dim<H, K>::HB;
};
// B operand of column block cb of a K-row activation tile [rows ; ones row ; zero padding]
// (explicitly GLOBAL pointers: a select between a kernel argument and the address of a __device__ object is a generic
//  pointer to the compiler, and flat loads return out of order -- every use would drain vmcnt(0))
template <int H, int K, int M, int METHOD>
__device__ __forceinline__ double duo_load_act(const DuoSrc<H, K, M, METHOD>& s, int jrow /* layer, M-1 = tanh */, int cb) {
  typedef ActLayout<H, K, M, RK<METHOD>::S> AL;
  const int lane = xw_lane(), j = lane & 3;
  const int row = 4 * cb + j;
  // lane j + 4 b + 16 k = (row 4 cb + j, path 4 k + b): lane-linear inside a full path-major block; a partial last block
  // of p rows holds (row, path) at p * path + row
  const int p = K - 4 * cb < 4 ? K - 4 * cb : 4;               // (compile-time after unrolling)
  const int off = p == 4 ? lane : p * (lane >> 2) + j;
  const xw_gptr src = row < K ? (xw_gptr)(s.A + (s.i * AL::STAGE + jrow * K + 4 * cb) * 16 + off) : (xw_gptr)&xw_duo_const[row == K ? 1 : 0][0];
  return __builtin_nontemporal_load(src);
}
// ... of [y_in ; t ; zero padding]: rows of y_l (checkpoints, stage 0) or of the activation record (later stages);
// branch-free (the loop must stay one basic block, or the compiler's wait-count bookkeeping falls back to vmcnt(0) at
// its head); the time row is patched in where the block is USED (a select here would wait for the load)
template <int H, int K, int M, int METHOD>
__device__ __forceinline__ double duo_load_y(const DuoSrc<H, K, M, METHOD>& s, int cb, int N, int tile) {
  typedef ActLayout<H, K, M, RK<METHOD>::S> AL;
  static_assert(H % 4 == 0, "the stage inputs are whole path-major blocks");
  const int lane = xw_lane(), j = lane & 3, b = (lane >> 2) & 3, k = lane >> 4;
  const int row = 4 * cb + j;
  const bool first = s.i == 0;                                // (wave-uniform)
  // stage 0: the checkpoint y_l [H][N] (row-major); later stages: path-major blocks of the record, lane-linear
  const long col = (long)tile * 16 + 4 * k + b;
  const double* __restrict__ src_y = s.Yl + (long)row * N + (col < N - 1 ? col : N - 1);
  const double* __restrict__ src_a = s.A + (long)(AL::YI + (s.i > 0 ? s.i - 1 : 0) * H + 4 * cb) * 16 + lane;
  const xw_gptr src = row < H ? (xw_gptr)(first ? src_y : src_a) : (xw_gptr)&xw_duo_const[0][0];
  return __builtin_nontemporal_load(src);
}
// A operand: rows 4 rb .. 4 rb + 3 of a transposed cotangent tile in LDS, posted by xw_writeT_pn (tile[row * XW_TSTRIDE +
// 4 (path & 3) + (path >> 2)]): lane i + 4 b + 16 k = (row i, path 4 k + b) sits at position 4 b + k
__device__ __forceinline__ double duo_readA(const double* tile, int rb) {
  const int l = xw_lane();
  return tile[(4 * rb + (l & 3)) * XW_TSTRIDE + ((l >> 2) & 3) * 4 + (l >> 4)];
}
// sum of an accumulator's four path-group partials (lane bits 2, 3), then element (row 4 rb + i, col 4 cb + j) from lane j + 16 i
__device__ __forceinline__ double duo_fold(double x) {
  x += __shfl_xor(x, 4);
  x += __shfl_xor(x, 8);
  return x;
}
template <int H, int K, int M, int METHOD>
__device__ __forceinline__ void duo_outer(const BwdJobs& jobs, const double* __restrict__ tf, const double* __restrict__ th,
                                          int L, int d, const double* qbuf, int vb) {
  typedef Dim<H, K> D;
  typedef RK<METHOD> T;
  typedef DuoPlan<H, K, M> P;
  typedef Duo4<H, K, M> Q;
  typedef DuoSrc<H, K, M, METHOD> Src;
  constexpr int NH = M > 1 ? M - 1 : 1;
  xw_setprio(jobs.prio);          // (a lower priority for this wave than for the chain: no difference)
  const int job = find_job(jobs, vb);
  const double* __restrict__ Y = jobs.Y[job];
  const double* __restrict__ act = jobs.act[job];
  const int N = jobs.N[job];
  const int tile = vb - jobs.tile0[job];
  const int lane = xw_lane();
  const UOff o = u_offsets(d, H, K);
  double gWh[Q::KB][Q::KB1], gWo[Q::HB][Q::KB1], gWy[Q::KB][Q::HB1];
#pragma unroll
  for (int rb = 0; rb < Q::KB; ++rb) {
#pragma unroll
    for (int cb = 0; cb < Q::KB1; ++cb) gWh[rb][cb] = 0.0;
#pragma unroll
    for (int cb = 0; cb < Q::HB1; ++cb) gWy[rb][cb] = 0.0;
  }
#pragma unroll
  for (int rb = 0; rb < Q::HB; ++rb)
#pragma unroll
    for (int cb = 0; cb < Q::KB1; ++cb) gWo[rb][cb] = 0.0;
  double Ra[Q::KB1], Rr[NH][Q::KB1], Ry[Q::HB1];          // B operands of the evaluation in flight
  const int E = (L - 1) * T::S;                            // field evaluations of the sweep
  double ti_cur = 0.0;                                     // time of the evaluation whose operands are in R*
  if (E > 0) {
    const Src s0(Y, act, tf, 0, L, N, tile);
    ti_cur = s0.ti;
#pragma unroll
    for (int cb = 0; cb < Q::KB1; ++cb) Ra[cb] = duo_load_act<H, K, M, METHOD>(s0, M - 1, cb);
#pragma unroll
    for (int jj = M - 2; jj >= 0; --jj)
#pragma unroll
      for (int cb = 0; cb < Q::KB1; ++cb) Rr[jj][cb] = duo_load_act<H, K, M, METHOD>(s0, jj, cb);
#pragma unroll
    for (int cb = 0; cb < Q::HB1; ++cb) Ry[cb] = duo_load_y<H, K, M, METHOD>(s0, cb, N, tile);
  }
  const bool trow = (lane & 3) == (H & 3);                 // lanes of the time row inside its column block H >> 2
  for (int e = 0; e < E; ++e) {
    // the chain wave has posted evaluation e (and is free to start e + 1).  No fence: an acquire would drain vmcnt and
    // with it the operand loads issued a whole evaluation ahead; LDS reads behind the barrier see the posted tiles.
    asm volatile("s_barrier" ::: "memory");
    const double* q = qbuf + (e & 1) * P::BUF;
    const Src sn(Y, act, tf, e + 1 < E ? e + 1 : e, L, N, tile);   // (the last evaluation reloads its own operands: no branch)
    // all A operands of the evaluation first (distinct registers, issued back to back: one exposed LDS latency per
    // evaluation), then block by block the matrix instructions and right behind them the reloads for the next evaluation
    double Ao[Q::HB], Az[M][Q::KB];
#pragma unroll
    for (int rb = 0; rb < Q::HB; ++rb) Ao[rb] = duo_readA(q + P::off(rb >> 2), rb & 3);
#pragma unroll
    for (int tq = 0; tq < M; ++tq)
#pragma unroll
      for (int rb = 0; rb < Q::KB; ++rb) Az[tq][rb] = duo_readA(q + P::off(D::HT + tq), rb);
    __builtin_amdgcn_sched_barrier(0);
    // cot(out) against [tanh ; 1]
#pragma unroll
    for (int cb = 0; cb < Q::KB1; ++cb)
#pragma unroll
      for (int rb = 0; rb < Q::HB; ++rb) gWo[rb][cb] = XW_MFMA4(Ao[rb], Ra[cb], gWo[rb][cb]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int cb = 0; cb < Q::KB1; ++cb) Ra[cb] = duo_load_act<H, K, M, METHOD>(sn, M - 1, cb);
    __builtin_amdgcn_sched_barrier(0);
    // cot(z_{j+1}) against [relu(z_j) ; 1], j = M-2 .. 0 (tile order of the chain wave)
#pragma unroll
    for (int jj = M - 2; jj >= 0; --jj) {
#pragma unroll
      for (int cb = 0; cb < Q::KB1; ++cb)
#pragma unroll
        for (int rb = 0; rb < Q::KB; ++rb) gWh[rb][cb] = XW_MFMA4(Az[M - 2 - jj][rb], Rr[jj][cb], gWh[rb][cb]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int cb = 0; cb < Q::KB1; ++cb) Rr[jj][cb] = duo_load_act<H, K, M, METHOD>(sn, jj, cb);
      __builtin_amdgcn_sched_barrier(0);
    }
    // cot(z_0) against [y_in ; t]: column H of dWy (the time row) is the time-column gradient
#pragma unroll
    for (int cb = 0; cb < Q::HB1; ++cb) {
      const double b = (cb == (H >> 2) && trow) ? ti_cur : Ry[cb];
#pragma unroll
      for (int rb = 0; rb < Q::KB; ++rb) gWy[rb][cb] = XW_MFMA4(Az[M - 1][rb], b, gWy[rb][cb]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int cb = 0; cb < Q::HB1; ++cb) Ry[cb] = duo_load_y<H, K, M, METHOD>(sn, cb, N, tile);
    __builtin_amdgcn_sched_barrier(0);
    ti_cur = sn.ti;
  }
  // ---- fold the path groups and store this tile's slab pieces: lane j + 16 i of block (rb, cb) holds (4 rb + i, 4 cb + j)
  double* slab = jobs.gslab[job] + (long)tile * o.total;
  const int i = lane >> 4, j = lane & 3;
  const bool owner = ((lane >> 2) & 3) == 0;
#pragma unroll
  for (int rb = 0; rb < Q::KB; ++rb)
#pragma unroll
    for (int cb = 0; cb < Q::KB1; ++cb) {
      const double x = duo_fold(gWh[rb][cb]);
      const int row = 4 * rb + i, col = 4 * cb + j;
      if (owner && row < K) {
        if (col < K) slab[o.Wh + row * K + col] = x;
        else if (col == K) slab[o.Whb + row] = x;
      }
    }
#pragma unroll
  for (int rb = 0; rb < Q::HB; ++rb)
#pragma unroll
    for (int cb = 0; cb < Q::KB1; ++cb) {
      const double x = duo_fold(gWo[rb][cb]);
      const int row = 4 * rb + i, col = 4 * cb + j;
      if (owner && row < H) {
        if (col < K) slab[o.Wo + row * K + col] = x;
        else if (col == K) slab[o.Wob + row] = x;
      }
    }
#pragma unroll
  for (int rb = 0; rb < Q::KB; ++rb)
#pragma unroll
    for (int cb = 0; cb < Q::HB1; ++cb) {
      const double x = duo_fold(gWy[rb][cb]);
      const int row = 4 * rb + i, col = 4 * cb + j;
      if (owner && row < K) {
        if (col < H) slab[o.Win + row * o.ldin + d + 1 + col] = x;
        else if (col == H) slab[o.Win + row * o.ldin + d] = x;
      }
    }
}

#else    // XW_ODE_WIDE16
// ---- the duo sweep's second wave in the wide container: weight gradients of the field on v_mfma_f64_16x16x4 --------------------
// One wave that runs the adjoint chain AND its 15 outer products per evaluation paid an LDS round trip per product in the middle
// of the chain and spilled 330 registers (641 us per sweep at the headline sample against 161 us without weight gradients).  As in
// the narrow containers the chain wave only POSTS its cotangent tiles (field_vjp OUTER = 2: cot(out) x HT, cot(z_{j+1}) of every
// tied layer, cot(z_0); two alternating buffers, one s_barrier per evaluation) and this wave, one evaluation behind, contracts
// them over the 16 paths with the layer inputs it loads from the activation store / the checkpoints itself, a whole evaluation
// ahead: 4 (HT + M - 1 + HT) matrix instructions per evaluation.  A operand = xw_readT of a posted tile (row i, path 4 ks + kk);
// B operand = (row j, path 4 ks + kk) of a 16-row block of the record, whose 4-row blocks are path-major (act_store): double
// 64 (j >> 2) + 4 (4 ks + kk) + (j & 3) of the block.  The bias gradients and the time column are row sums of the posted tiles: a
// lane adds the A operands it reads anyway, the four lane groups are folded once at the end.
template <int H, int K, int M, int METHOD>
__device__ __forceinline__ void duo_outer(const BwdJobs& jobs, const double* __restrict__ tf, const double* __restrict__ th,
                                          int L, int d, const double* qbuf, int vb) {
  typedef Dim<H, K> D;
  typedef RK<METHOD> T;
  typedef DuoPlan<H, K, M> P;
  typedef ActLayout<H, K, M, T::S> AL;
  constexpr int NH = M > 1 ? M - 1 : 1;
  xw_setprio(jobs.prio);
  const int job = find_job(jobs, vb);
  const double* __restrict__ Y = jobs.Y[job];
  const double* __restrict__ act = jobs.act[job];
  const int N = jobs.N[job];
  const int tile = vb - jobs.tile0[job];
  const int lane = xw_lane(), j = lane & 15, kk = lane >> 4;
  const UOff o = u_offsets(d, H, K);
  const int lo = 64 * (j >> 2) + 4 * kk + (j & 3);
  const long ntile = (N + 15) >> 4;
  long ycol[4];                                          // columns of the checkpoint this lane reads (clamped: the last tile's padding paths)
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const long c = (long)tile * 16 + 4 * ks + kk;
    ycol[ks] = c < N ? c : N - 1;
  }
  d4 gWo[D::HT], gWy[D::HT], gWh = xw_zero4();
  double sbo[D::HT], sbh = 0.0, swt = 0.0;
#pragma unroll
  for (int ht = 0; ht < D::HT; ++ht) {
    gWo[ht] = xw_zero4();
    gWy[ht] = xw_zero4();
    sbo[ht] = 0.0;
  }
  double Ra[4], Rr[NH][4], Ry[D::HT][4];                 // B operands of the evaluation in flight
  const int E = (L - 1) * T::S;                          // field evaluations of the sweep (chain-wave order: steps L-2 .. 0, stages S-1 .. 0)
  // operands of evaluation e: the record of its step, its stage, its time
  auto load_eval = [&](int e, double& ti) {
    const int l = L - 2 - e / T::S, i = T::S - 1 - e % T::S;
    const double* __restrict__ A = act + ((long)l * ntile + tile) * (AL::TOTAL * 16) + lo;
    const double t0 = tf[l];
    ti = t0 + T::c(i) * (tf[l + 1] - t0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) Ra[ks] = __builtin_nontemporal_load(A + (i * AL::STAGE + (M - 1) * K) * 16 + 16 * ks);
#pragma unroll
    for (int jj = 0; jj < M - 1; ++jj)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) Rr[jj][ks] = __builtin_nontemporal_load(A + (i * AL::STAGE + jj * K) * 16 + 16 * ks);
    // the field's input: the checkpoint y_l [H][N] (stage 0) or the stage input kept in the record
    const bool first = i == 0;                           // (wave-uniform)
#pragma unroll
    for (int ct = 0; ct < D::HT; ++ct)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const double* __restrict__ sy = Y + ((long)l * H + 16 * ct + j) * N + ycol[ks];
        const double* __restrict__ sa = A + (long)(AL::YI + (i > 0 ? i - 1 : 0) * H + 16 * ct) * 16 + 16 * ks;
        Ry[ct][ks] = __builtin_nontemporal_load(first ? sy : sa);
      }
  };
  double ti_cur = 0.0;
  if (E > 0) load_eval(0, ti_cur);
  for (int e = 0; e < E; ++e) {
    // the chain wave has posted evaluation e (and is free to start e + 1).  No fence: an acquire would drain vmcnt and with it the
    // operand loads issued a whole evaluation ahead; LDS reads behind the barrier see the posted tiles.
    asm volatile("s_barrier" ::: "memory");
    const double* q = qbuf + (e & 1) * P::BUF;
    double Ao[D::HT][4], Az[M][4];
#pragma unroll
    for (int ht = 0; ht < D::HT; ++ht)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) Ao[ht][ks] = xw_readT(q + P::off(ht), ks);
#pragma unroll
    for (int tq = 0; tq < M; ++tq)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) Az[tq][ks] = xw_readT(q + P::off(D::HT + tq), ks);
    __builtin_amdgcn_sched_barrier(0);
    // cot(out) against tanh(z_{m-1})
#pragma unroll
    for (int ht = 0; ht < D::HT; ++ht) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) gWo[ht] = XW_MFMA(Ao[ht][ks], Ra[ks], gWo[ht]);
      sbo[ht] += (Ao[ht][0] + Ao[ht][1]) + (Ao[ht][2] + Ao[ht][3]);
    }
    // cot(z_{j+1}) against relu(z_j), j = M-2 .. 0 (tile order of the chain wave); the record keeps the layer INPUT z_j
#pragma unroll
    for (int jj = M - 2; jj >= 0; --jj) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const double r = Rr[jj][ks];
        gWh = XW_MFMA(Az[M - 2 - jj][ks], r, gWh);
      }
      sbh += (Az[M - 2 - jj][0] + Az[M - 2 - jj][1]) + (Az[M - 2 - jj][2] + Az[M - 2 - jj][3]);
    }
    // cot(z_0) against the field's input; its row sums times t are the time column
#pragma unroll
    for (int ct = 0; ct < D::HT; ++ct)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) gWy[ct] = XW_MFMA(Az[M - 1][ks], Ry[ct][ks], gWy[ct]);
    swt = fma(ti_cur, (Az[M - 1][0] + Az[M - 1][1]) + (Az[M - 1][2] + Az[M - 1][3]), swt);
    __builtin_amdgcn_sched_barrier(0);
    load_eval(e + 1 < E ? e + 1 : e, ti_cur);            // (the last evaluation reloads its own operands: no branch)
    __builtin_amdgcn_sched_barrier(0);
  }
  // ---- this tile's slab pieces: accumulator register r of lane (g, n) = element (row g + 4 r, column n)
  double* slab = jobs.gslab[job] + (long)tile * o.total;
  storeD(slab + o.Wh, K, K, K, 0, 0, gWh);                  // (u_layers = 1: zeros -- the slot of the missing tied layer)
#pragma unroll
  for (int ct = 0; ct < D::HT; ++ct) storeD(slab + o.Win + d + 1, o.ldin, K, H, 0, 16 * ct, gWy[ct]);
#pragma unroll
  for (int ht = 0; ht < D::HT; ++ht) storeD(slab + o.Wo, K, H, K, 16 * ht, 0, gWo[ht]);
  // row sums: lane (row j, group kk) holds its group's share
  auto fold = [](double x) {
    x += __shfl_xor(x, 16);
    x += __shfl_xor(x, 32);
    return x;
  };
  sbh = fold(sbh);
  swt = fold(swt);
#pragma unroll
  for (int ht = 0; ht < D::HT; ++ht) sbo[ht] = fold(sbo[ht]);
  if (kk == 0) {
    if (j < K) slab[o.Whb + j] = sbh;
    if (j < K) slab[o.Win + (long)j * o.ldin + d] = swt;
#pragma unroll
    for (int ht = 0; ht < D::HT; ++ht)
      if (16 * ht + j < H) slab[o.Wob + 16 * ht + j] = sbo[ht];
  }
}

#endif   // XW_ODE_WIDE16
template <int H, int K, int M, int METHOD, bool PARAMS, bool SAVED, bool ADJ = false>
__global__ void __launch_bounds__(64) k_ode_bwd(const BwdJobs jobs, const double* __restrict__ tf,
                                                const double* __restrict__ th, int L, int d) {
  __shared__ double lds[XW_SWEEP_TILES * XW_TTILE];   // (plan: XW_SWEEP_TILES)
  sweep_body<H, K, M, METHOD, PARAMS, SAVED, ADJ, false>(jobs, tf, th, L, d, lds, nullptr, (int)blockIdx.x);
}
// the duo sweep: wave 0 = adjoint chain, wave 1 = weight gradients of the field (see sweep_body / duo_outer).
// Placement (tools/probe_place.hip, profiles/r02_probe_place.txt): the dispatcher puts a block's waves on consecutive
// SIMDs and starts the next block of the same CU ONE SIMD further, so two of these blocks on a CU land on SIMDs (0,1),
// (1,2).  Both remedies were built and measured at 1 / 2 / 3 concurrent jobs of 4096 paths (this form: 105 / 147 / 170 us):
// a spacer wave that ends at once (blocks on (0,2), (1,3); but a block then needs three waves' registers to start, two
// blocks per CU instead of four): 105 / 142 / 181 us, generator sub-step 0.572 ms against 0.518; two tiles per four-wave
// block (every SIMD of the CU, tiles coupled through the block barrier, a job on half the CUs): 118 / 126 / 216 us,
// generator sub-step 0.533 ms.  Two tiles on one CU slow each other by a quarter wherever their waves sit.
#define XW_DUO_THREADS 128
template <int H, int K, int M, int METHOD>
__global__ void __launch_bounds__(XW_DUO_THREADS) k_ode_bwd_duo(const BwdJobs jobs, const double* __restrict__ tf,
                                                                const double* __restrict__ th, int L, int d) {
  __shared__ double lds[2 * DuoPlan<H, K, M>::BUF];
  // jobs.spread (= the chip's CU count, 0: off): every second ROUND of blocks is a spacer that ends at once.  The dispatcher deals
  // blocks round-robin over the CUs and starts a CU's next block one SIMD further than the last (profiles/r02_probe_place.txt):
  // two of these two-wave blocks on a CU sit on SIMDs (0, 1) and (1, 2) -- one SIMD hosts a chain wave AND a partner wave, one
  // stands idle, and a launch of 512 tiles takes 142 us where 256 take 92.  With a spacer round in between the second block
  // starts on SIMD 2: (0, 1), (2, 3).
  int vb = (int)blockIdx.x;
  if (jobs.spread > 0) {
    const int round = vb / jobs.spread;
    if (round & 1) return;
    vb -= (round >> 1) * jobs.spread;
    if (vb >= jobs.tile0[jobs.n]) return;
  }
  if (threadIdx.x < 64) sweep_body<H, K, M, METHOD, true, true, false, true>(jobs, tf, th, L, d, nullptr, lds, vb);
  else duo_outer<H, K, M, METHOD>(jobs, tf, th, L, d, lds, vb);
}

#ifndef XW_ODE_WIDE16
#include "xw_ode_n4.h"
#endif   // (no narrow tiles in the wide container)

template <int H, int K, int M>
int launch_fwd(int method, const FwdJobs& jobs, const double* t, const double* theta, int L, int d, hipStream_t s) {
  const dim3 grid(jobs.tile0[jobs.n]), block(64);
  bool act = true;                                     // all jobs or none, all in the same mode (checked by the caller)
  for (int i = 0; i < jobs.n; ++i) act = act && jobs.act[i] != nullptr;
#ifndef XW_ODE_WIDE16
  if (jobs.narrow) {
    // narrow tiles (xw_ode_n4.h): the same grid of 16-path tiles, four waves of 4 paths each
    switch (method * 3 + (act ? (jobs.x_only ? 2 : 1) : 0)) {
      case 0: hipLaunchKernelGGL((n4::k_ode_fwd_n4<H, K, M, 0, 0>), grid, dim3(256), 0, s, jobs, t, theta, L, d); break;
      case 1: hipLaunchKernelGGL((n4::k_ode_fwd_n4<H, K, M, 0, 1>), grid, dim3(256), 0, s, jobs, t, theta, L, d); break;
      case 2: hipLaunchKernelGGL((n4::k_ode_fwd_n4<H, K, M, 0, 2>), grid, dim3(256), 0, s, jobs, t, theta, L, d); break;
      case 3: hipLaunchKernelGGL((n4::k_ode_fwd_n4<H, K, M, 1, 0>), grid, dim3(256), 0, s, jobs, t, theta, L, d); break;
      case 4: hipLaunchKernelGGL((n4::k_ode_fwd_n4<H, K, M, 1, 1>), grid, dim3(256), 0, s, jobs, t, theta, L, d); break;
      case 5: hipLaunchKernelGGL((n4::k_ode_fwd_n4<H, K, M, 1, 2>), grid, dim3(256), 0, s, jobs, t, theta, L, d); break;
      case 6: hipLaunchKernelGGL((n4::k_ode_fwd_n4<H, K, M, 2, 0>), grid, dim3(256), 0, s, jobs, t, theta, L, d); break;
      default: return XW_E_ARG;
    }
    return xw_launch_status();
  }
#endif
  switch (method * 3 + (act ? (jobs.x_only ? 2 : 1) : 0)) {
    case 0: hipLaunchKernelGGL((k_ode_fwd<H, K, M, 0, 0>), grid, block, 0, s, jobs, t, theta, L, d); break;
    case 1: hipLaunchKernelGGL((k_ode_fwd<H, K, M, 0, 1>), grid, block, 0, s, jobs, t, theta, L, d); break;
    case 2: hipLaunchKernelGGL((k_ode_fwd<H, K, M, 0, 2>), grid, block, 0, s, jobs, t, theta, L, d); break;
    case 3: hipLaunchKernelGGL((k_ode_fwd<H, K, M, 1, 0>), grid, block, 0, s, jobs, t, theta, L, d); break;
    case 4: hipLaunchKernelGGL((k_ode_fwd<H, K, M, 1, 1>), grid, block, 0, s, jobs, t, theta, L, d); break;
    case 5: hipLaunchKernelGGL((k_ode_fwd<H, K, M, 1, 2>), grid, block, 0, s, jobs, t, theta, L, d); break;
    case 6: hipLaunchKernelGGL((k_ode_fwd<H, K, M, 2, 0>), grid, block, 0, s, jobs, t, theta, L, d); break;
    default: return XW_E_ARG;
  }
  return xw_launch_status();
}
template <int H, int K, int M, bool PARAMS>
int launch_bwd(int method, const BwdJobs& jobs, const double* t, const double* theta, int L, int d, bool adj, bool narrow,
               hipStream_t s) {
  const dim3 grid(jobs.tile0[jobs.n]), block(64);
#ifdef XW_ODE_WIDE16
  // the wide container: no narrow tiles; the recomputing sweeps form their weight gradients themselves (OUTER = 1)
  {
    bool act_ = true;
    for (int i = 0; i < jobs.n; ++i) act_ = act_ && jobs.act[i] != nullptr;
    if (adj || !act_ || method > 1)
      return XW_ODE_FN(xw_ode_bwd_recomp_w)(&jobs, t, theta, method, L, d, M, PARAMS ? 1 : 0, adj ? 1 : 0, (void*)s);
    // from the activation store: without weight gradients one wave per tile, with them the duo sweep (chain wave + partner wave)
    if (PARAMS) {
      BwdJobs jd = jobs;
      jd.spread = 0;
      if (method == 0) hipLaunchKernelGGL((k_ode_bwd_duo<H, K, M, 0>), grid, dim3(XW_DUO_THREADS), 0, s, jd, t, theta, L, d);
      else hipLaunchKernelGGL((k_ode_bwd_duo<H, K, M, 1>), grid, dim3(XW_DUO_THREADS), 0, s, jd, t, theta, L, d);
    } else {
      if (method == 0) hipLaunchKernelGGL((k_ode_bwd<H, K, M, 0, false, true>), grid, block, 0, s, jobs, t, theta, L, d);
      else hipLaunchKernelGGL((k_ode_bwd<H, K, M, 1, false, true>), grid, block, 0, s, jobs, t, theta, L, d);
    }
    return xw_launch_status();
  }
#else
  if (narrow) {
    // narrow tiles (xw_ode_n4.h): the same grid of 16-path tiles, four waves of 4 paths each; from the activation store only
    if (adj || method > 1) return XW_E_ARG;
    for (int i = 0; i < jobs.n; ++i)
      if (jobs.act[i] == nullptr) return XW_E_ARG;
    if (method == 0) hipLaunchKernelGGL((n4::k_ode_bwd_n4<H, K, M, 0, PARAMS>), grid, dim3(256), 0, s, jobs, t, theta, L, d);
    else hipLaunchKernelGGL((n4::k_ode_bwd_n4<H, K, M, 1, PARAMS>), grid, dim3(256), 0, s, jobs, t, theta, L, d);
    return xw_launch_status();
  }
  bool act = true;                                     // all jobs or none (checked by the caller)
  for (int i = 0; i < jobs.n; ++i) act = act && jobs.act[i] != nullptr;
  if (adj || !act || method > 1)      // (the recomputing sweeps live in an object of their own: XW_ODE_PART_RECOMP below)
    return XW_ODE_FN(xw_ode_bwd_recomp_w)(&jobs, t, theta, method, L, d, M, PARAMS ? 1 : 0, adj ? 1 : 0, (void*)s);
  // two rounds of tiles over the CUs: a spacer round in between (k_ode_bwd_duo)
  static const int spread_on = [] { const char* e = getenv("XW_DUO_SPREAD"); return e ? atoi(e) : 1; }();
  static const int ncu = [] { int dev = 0, n = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0; return n; }();
  const int tiles = jobs.tile0[jobs.n];
  BwdJobs jd = jobs;
  // (exactly two rounds: 103 against 142 us at 512 tiles; with three rounds the third block of a CU meets the first again either
  //  way -- 149 against 145 us --, and from four rounds on every SIMD hosts two waves whatever the order)
  jd.spread = (spread_on && ncu > 0 && tiles > ncu && tiles <= 2 * ncu) ? ncu : 0;
  const int rounds = jd.spread ? (tiles + ncu - 1) / ncu : 0;
  const dim3 duo_grid(jd.spread ? (2 * (rounds - 1)) * ncu + (tiles - (rounds - 1) * ncu) : tiles);
  switch (method) {
    case 0:
      if (PARAMS) hipLaunchKernelGGL((k_ode_bwd_duo<H, K, M, 0>), duo_grid, dim3(XW_DUO_THREADS), 0, s, jd, t, theta, L, d);
      else hipLaunchKernelGGL((k_ode_bwd<H, K, M, 0, false, true>), grid, block, 0, s, jobs, t, theta, L, d);
      break;
    case 1:
      if (PARAMS) hipLaunchKernelGGL((k_ode_bwd_duo<H, K, M, 1>), duo_grid, dim3(XW_DUO_THREADS), 0, s, jd, t, theta, L, d);
      else hipLaunchKernelGGL((k_ode_bwd<H, K, M, 1, false, true>), grid, block, 0, s, jobs, t, theta, L, d);
      break;
    default: return XW_E_ARG;
  }
  return xw_launch_status();
#endif
}

}  // namespace

#ifndef XW_ODE_PART_RECOMP
extern "C" int XW_ODE_FN(xw_ode_fwd_multi_w)(const XwOdeFwdJob* jobs, int njobs, const double* t, const double* theta, int method,
                                             int L, int d, int m, double* zero16, void* stream) {
  if (!jobs || njobs < 1 || njobs > XW_MAXJOBS || !t || !theta || L <= 0 || d <= 0 || m < 1) return XW_E_ARG;
  FwdJobs J;
  J.n = njobs;
  J.zero16 = zero16;
  J.x_only = jobs[0].act_x_only ? 1 : 0;
  J.narrow = jobs[0].narrow ? 1 : 0;
  J.prio = XW_ODE_PRIO - (jobs[0].prio_drop < 0 ? 0 : jobs[0].prio_drop > 3 ? 3 : jobs[0].prio_drop);
  J.tile0[0] = 0;
  for (int i = 0; i < XW_MAXJOBS; ++i) {
    const bool on = i < njobs;
    if (on && (!jobs[i].xT || !jobs[i].start || !jobs[i].u || jobs[i].N <= 0)) return XW_E_ARG;
    if (on && (jobs[i].act_x_only ? 1 : 0) != J.x_only) return XW_E_ARG;         // one store mode per launch
    if (on && (jobs[i].narrow ? 1 : 0) != J.narrow) return XW_E_ARG;              // one layout per launch
    J.xT[i] = on ? jobs[i].xT : nullptr;
    J.start[i] = on ? jobs[i].start : nullptr;
    J.u[i] = on ? jobs[i].u : nullptr;
    J.Y[i] = on ? jobs[i].Y : nullptr;
    J.act[i] = (on && method != 2) ? jobs[i].act : nullptr;     // (rk4 sweeps recompute: nothing to store)
    if (on && (J.act[i] != nullptr) != (J.act[0] != nullptr)) return XW_E_ARG;   // all groups of a launch, or none
    J.N[i] = on ? jobs[i].N : 0;
    J.tile0[i + 1] = J.tile0[i] + (on ? (jobs[i].N + 15) / 16 : 0);
  }
  hipStream_t s = (hipStream_t)stream;
#define CALL(HH, KK, MM) return launch_fwd<HH, KK, MM>(method, J, t, theta, L, d, s);
  XW_ODE_DISPATCH(CALL)
#undef CALL
}

extern "C" int XW_ODE_FN(xw_ode_bwd_multi_w)(const XwOdeBwdJob* jobs, int njobs, const double* t, const double* theta, int method,
                                             int L, int d, int m, int mode, void* stream) {
  if (!jobs || njobs < 1 || njobs > XW_MAXJOBS || !t || !theta || L <= 0 || d <= 0 || m < 1 || (mode & 3) == 0 || ((mode & 4) && (mode & 3) != 3) || ((mode & 8) && (mode & 4)) || ((mode & 16) && (mode & 8)) || (mode & ~127)) return XW_E_ARG;
  BwdJobs J;
  J.n = njobs;
  J.x_ones = (mode & 4) ? 1 : 0;
  J.spread = 0;
  J.prio = XW_ODE_PRIO - ((mode >> 5) & 3);
  J.tile0[0] = 0;
  for (int i = 0; i < XW_MAXJOBS; ++i) {
    const bool on = i < njobs;
    if (on) {
      if (!jobs[i].xT || !jobs[i].start || !jobs[i].Y || jobs[i].N <= 0) return XW_E_ARG;
      if ((mode & 2) && !jobs[i].gslab) return XW_E_ARG;
      if ((mode & 1) && !(mode & 4) && (!jobs[i].gx || !jobs[i].gs)) return XW_E_ARG;
      if ((mode & 1) && (!jobs[i].gx != !jobs[i].gs)) return XW_E_ARG;
    }
    J.xT[i] = on ? jobs[i].xT : nullptr;
    J.start[i] = on ? jobs[i].start : nullptr;
    J.Y[i] = on ? jobs[i].Y : nullptr;
    J.act[i] = (on && method != 2 && !(mode & 8)) ? jobs[i].act : nullptr;   // (rk4 and the continuous adjoint recompute)
    if (on && (J.act[i] != nullptr) != (J.act[0] != nullptr)) return XW_E_ARG;   // all groups of a launch, or none
    J.ubar[i] = on ? jobs[i].ubar : nullptr;
    J.res_u[i] = on ? jobs[i].res_u : nullptr;
    J.res_ref[i] = on ? jobs[i].res_ref : nullptr;
    J.res_coef[i] = on ? jobs[i].res_coef : 0.0;
    J.res_base[i] = on ? jobs[i].res_base : 0.0;
    J.res_first[i] = on ? jobs[i].res_first_only : 0;
    J.res_w[i] = on ? jobs[i].res_w : nullptr;
    J.res_c[i] = on ? jobs[i].res_c : nullptr;
    J.res_cp[i] = on ? jobs[i].res_cp : nullptr;
    J.res_kappa2[i] = on ? jobs[i].res_kappa2 : 0.0;
    J.res_wpp[i] = on ? jobs[i].res_w_per_point : 0;
    if (on && jobs[i].res_u && (!jobs[i].res_ref || jobs[i].ubar)) return XW_E_ARG;
    if (on && jobs[i].res_u && jobs[i].res_first_only == 2 && (!jobs[i].res_w || (!jobs[i].res_c != !jobs[i].res_cp))) return XW_E_ARG;
    if (on && (jobs[i].res_first_only < 0 || jobs[i].res_first_only > 2)) return XW_E_ARG;
    J.gx[i] = (on && (mode & 1)) ? jobs[i].gx : nullptr;
    J.gs[i] = (on && (mode & 1)) ? jobs[i].gs : nullptr;
    J.gslab[i] = (on && (mode & 2)) ? jobs[i].gslab : nullptr;
    J.N[i] = on ? jobs[i].N : 0;
    J.tile0[i + 1] = J.tile0[i] + (on ? (jobs[i].N + 15) / 16 : 0);
  }
  hipStream_t s = (hipStream_t)stream;
#define CALL(HH, KK, MM)                                                             \
  return (mode & 2) ? launch_bwd<HH, KK, MM, true>(method, J, t, theta, L, d, (mode & 8) != 0, (mode & 16) != 0, s)     \
                    : launch_bwd<HH, KK, MM, false>(method, J, t, theta, L, d, (mode & 8) != 0, (mode & 16) != 0, s);
  XW_ODE_DISPATCH(CALL)
#undef CALL
}
#else   // XW_ODE_PART_RECOMP: only the sweeps that re-evaluate the field (no activation store) and the continuous adjoint
extern "C" int XW_ODE_FN(xw_ode_bwd_recomp_w)(const void* jobs_, const double* t, const double* theta, int method, int L, int d,
                                              int m, int params, int adj, void* stream) {
  const BwdJobs& jobs = *static_cast<const BwdJobs*>(jobs_);
  const dim3 grid(jobs.tile0[jobs.n]), block(64);
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(HH, KK, MM, PP, AA)                                                                                                 \
  switch (method) {                                                                                                                \
    case 0: hipLaunchKernelGGL((k_ode_bwd<HH, KK, MM, 0, PP, false, AA>), grid, block, 0, s, jobs, t, theta, L, d); break;         \
    case 1: hipLaunchKernelGGL((k_ode_bwd<HH, KK, MM, 1, PP, false, AA>), grid, block, 0, s, jobs, t, theta, L, d); break;         \
    case 2: hipLaunchKernelGGL((k_ode_bwd<HH, KK, MM, 2, PP, false, AA>), grid, block, 0, s, jobs, t, theta, L, d); break;         \
    default: return XW_E_ARG;                                                                                                      \
  }                                                                                                                                \
  return xw_launch_status();
#define CALL(HH, KK, MM)                                                                                                           \
  if (params && adj) { LAUNCH(HH, KK, MM, true, true) }                                                                            \
  else if (params) { LAUNCH(HH, KK, MM, true, false) }                                                                             \
  else if (adj) { LAUNCH(HH, KK, MM, false, true) }                                                                                \
  else { LAUNCH(HH, KK, MM, false, false) }
  XW_ODE_DISPATCH(CALL)
#undef CALL
#undef LAUNCH
}
#endif
