// xw_disc.hip -- adversarial test network v_phi on gfx950.
//
// Replaces discriminator.forward (src/model.py:37-47 of the reference), the helper backward that produces
// nabla phi (src/loss.py:60-63) and the autograd replay that produces phi's gradient (src/training.py:161).
//
// v_phi is a W-wide (50) MLP with ONE weight-tied hidden layer applied q (9) times: a real dense contraction
// [W x W] x [W x points] -- the only part of the hot path where the hidden width makes MFMA tiles pay
// (4 row tiles x 13 k-steps of v_mfma_f64_16x16x4_f64 per layer per 16 points, 75 % useful after padding 50 -> 64/52).
//
//   k_disc_fwd : value and d/dt (forward-mode tangent: the weak form needs d(phi)/dt at every sample point but
//                nabla_x phi only at the initial time).  Vh rows 0..47 as 39 A-fragments in LDS
//                (shared by the 4 waves of a block, two waves per SIMD), rows 48, 49 on the vector ALU; activations never
//                leave the chain layout.  HBM traffic: x, t in, v, dv/dt out -- plus, on request, the record of layer
//                inputs (500 doubles per point) that k_disc_rec reads instead of recomputing the forward.
//   k_disc_rec : parameter gradient from that record: reverse chain + MFMA outer products (48 x 48 core on 16x16x4 tiles, the
//                two edges of the 50 x 51 matrix on 4x4x4 blocks), 78 KB of LDS, two blocks/CU.
//   k_disc_bwd : recomputes the forward for a tile of 16 points per wave (activations stay in registers), runs the
//                reverse chain with Vh^T fragments from LDS, and accumulates  dVh += delta_{j+1} (x) relu(a_j)  as MFMA
//                outer products over the points (LDS transpose; 4 waves of a block own one 16-row band of dVh each).
//                Optionally also returns the input gradient (nabla_x v, dv/dt) -- used for nabla phi at t0.
#include "xw_common.h"
#include "xw_generic.h"
#include <atomic>
#include <cstdlib>

namespace {

template <int W> struct VDim {
  static constexpr int MT = (W + 15) / 16;
  static constexpr int KS = (W + 3) / 4;
  // rows of the last (partial) row tile.  A 16-row MFMA tile for 2 live rows (W = 50) is 13 x 2 wasted 64-cycle matrix
  // instructions per layer: when the tail is short the forward kernel contracts those rows on the vector ALU instead.
  static constexpr int TR = W - 16 * (MT - 1);
  static constexpr bool VTAIL = TR <= 3;
  static constexpr int MTF = VTAIL ? MT - 1 : MT;     // row tiles that go through the matrix pipe in k_disc_fwd
  // live registers (4-row groups) of row tile mt, counting the ones row W of the outer products: elementwise work,
  // checkpoints and transposes skip the padding rows (a lone wave pays ~9 clocks per FP64 VALU instruction)
  __device__ static constexpr int LR(int mt) { return (W + 1 - 16 * mt) >= 16 ? 4 : (W + 1 - 16 * mt + 3) / 4; }
  // W not a multiple of 16: a padding row of the last tile is set to one, so that column W of the dVh outer products
  // collects dVh.b for free; otherwise (W = 64) the bias gradient is summed on the vector ALU
  static constexpr bool BIASROW = (W % 16) != 0;
  // (MT = 4: widths 50 and 64, every kernel of this file; MT = 6, 8: the 96- and 128-wide containers -- forward and reverse from the record)
  static_assert(MT == 4 || MT == 6 || MT == 8, "row tiles of the compiled widths");
};

// point -> (time, path) ; path mode: p = l*N + n ; point mode (tpp != null): p = n, L == 1
struct Pt {
  int p, n;
  bool valid;
  double t;
};
__device__ __forceinline__ Pt locate(long tile, long P, int N, const double* __restrict__ tf, const double* __restrict__ tpp) {
  Pt q;
  const long p = tile * 16 + (xw_lane() & 15);
  q.valid = p < P;
  const long pc = q.valid ? p : P - 1;
  q.p = (int)pc;
  if (tpp != nullptr) {
    q.n = (int)pc;
    q.t = tpp[pc];
  } else {
    const int l = (int)(pc / N);
    q.n = (int)(pc - (long)l * N);
    q.t = tf[l];
  }
  return q;
}

// input layer a0 = Vin [t; x] + b  (and its t-tangent = Vin[:, 0])
template <int W>
__device__ __forceinline__ void input_layer(const double* __restrict__ ph, const VOff& o, const double* __restrict__ xT,
                                            int N, int d, const Pt& q, d4 (&a)[VDim<W>::MT], d4 (&ad)[VDim<W>::MT]) {
  typedef VDim<W> D;
  const int g = xw_lane() >> 4;
#pragma unroll
  for (int mt = 0; mt < D::MT; ++mt) {
    ad[mt] = xw_vecD_strided(ph + o.Vin, o.ldin, W, 16 * mt);
    a[mt] = xw_vecD(ph + o.Vinb, W, 16 * mt) + ad[mt] * q.t;
  }
  for (int ks = 0; ks < (d + 3) / 4; ++ks) {
    const int i = 4 * ks + g;
    const double b = i < d ? xT[(long)i * N + q.n] : 0.0;
#pragma unroll
    for (int mt = 0; mt < D::MT; ++mt) a[mt] = XW_MFMA(xw_fragA(ph + o.Vin + 1, o.ldin, W, d, 16 * mt, 4 * ks), b, a[mt]);
  }
}

// ------------------------------------------------------------------------------------------------------------------
#define XW_QMAX 16   // deepest test network whose ReLU masks fit the LDS stash of k_disc_fwd's fused input gradient
#define XW_VIN_KS 6  // k-steps of the input layer (d <= 24) whose A-fragments k_disc_fwd keeps in LDS (44 KB per block in all:
#define XW_VIN_KS_WIDE 13  // second instantiation: d <= 52 entirely from LDS (57 KB per block), wider inputs (d = 100: 25 k-steps)
                           // take their first 13 k-steps from LDS and the rest from global memory
                     // two of its blocks and two blocks of the duo sweep still share a CU's 160 KB)

// DYN: only a wave's first tile is its static one; every later tile comes from ticket counters (one returning atomic per
// tile by lane 0, issued before the input layer and read after it).  A SIMD's vector and FP64 matrix instructions add up
// across all waves it hosts (profiles/r02_probe_coexec.txt), and with 3/4 of the block slots every second CU hosts two
// blocks of this kernel and the others one block plus the stepper's waves: a static split ends with the most loaded SIMD.
// The waves of a launch move in step, so their fetches come in bursts and one word serves only ~88 atomics/us: the tiles
// after the first round are dealt round-robin to XW_DISC_NQ sub-queues (one 256-byte line each), a block's home queue is
// blockIdx mod NQ, and a wave whose queue ran dry tries the next XW_DISC_STEAL ones before it stops.  The last wave out
// (LDS count per block, then one global count) zeroes the words for the slot's next launch.
#define XW_DISC_NQ 32
#define XW_DISC_STEAL 3
#define XW_DISC_QSTRIDE 64
#define XW_DISC_SLOTS 128
__device__ unsigned int xw_disc_queue[XW_DISC_SLOTS][(XW_DISC_NQ + 1) * XW_DISC_QSTRIDE];
#ifdef XW_CLOCK_PROBE   // diagnostic build only (tools/probe_disc_clock.py): shader clocks / 100 MHz ticks of every wave's tile loop
__device__ unsigned long long xw_clock_buf[2 * 4096];
#endif
#ifndef XW_DISC_FWD_WAVES
#define XW_DISC_FWD_WAVES 2     // waves per SIMD the register allocation leaves room for (2: 256 registers, 3: 168)
#endif
// VKS: k-steps of the input layer kept in LDS (0: none; -1: the x-projection Vin[:, 1..d] x + Vin.b of every PATH is handed in,
// `xproj`, and the input layer of a point is one load and one multiply-add per row: on vertical paths the d columns of the input
// layer do not move along a path -- L = 32..64 times the same d W multiply-adds per path, 12 % of the launch's matrix
// instructions at d = 100 and 25 + 48 loads per tile)
template <int W, bool ACT, bool DYN, int VKS>
__global__ void __launch_bounds__(256, (W > 64 ? 1 : XW_DISC_FWD_WAVES)) k_disc_fwd(const double* __restrict__ xT, const double* __restrict__ tf,
                                                     const double* __restrict__ tpp, const double* __restrict__ ph, int N,
                                                     int L, int d, int q, double* __restrict__ v, double* __restrict__ vt,
                                                     double* __restrict__ gxv, double* __restrict__ gtv, int ngrad,
                                                     double* __restrict__ act, unsigned int* __restrict__ queue,
                                                     const double* __restrict__ xproj) {
  typedef VDim<W> D;
  // Vh as MFMA A-fragments in LDS (26.6 KB), shared by the 4 waves of the block: one ds_read_b64 feeds two 64-cycle MFMAs
  // (value and d/dt tangent), and keeping them out of the register file lets two waves share a SIMD so that one wave's
  // relu / tanh VALU work overlaps the other's matrix work.
  __shared__ double sVh[D::MTF * D::KS * 64];
  __shared__ double sB[2 * 16 * D::MT];
  __shared__ double sT[D::KS * 4 * (D::VTAIL ? D::TR : 1)];   // Vh[16 (MT-1) + r][4 ks + g]: the tail rows, per lane group
  __shared__ unsigned int sMask[XW_QMAX][256];     // ReLU masks (4 MT rows per lane) of the layers, for the fused reverse chain
  // input layer: Vin[:, 1..d] as A-fragments, Vin[:, 0] and Vin.b as row vectors.  From global memory they were 52 loads per
  // tile (the fragment loads put the four lanes of a quad into four rows) and their address arithmetic, re-formed for
  // every tile to keep them out of the spilled loop-invariant set
  constexpr bool VINLDS = VKS > 0;
  constexpr bool XPROJ = VKS < 0;
  __shared__ double sVin[VINLDS ? VKS * D::MT * 64 : 1];
  __shared__ double sIn[(VINLDS || XPROJ) ? 2 * 16 * D::MT : 1];
  __shared__ unsigned int sDone;
  if (DYN && threadIdx.x == 0) sDone = 0;
  // (s_setprio for this kernel's waves: no gain at 1, -5 % at 3 -- the sub-step's SIMD time is conserved)
  const int lane = xw_lane(), g = lane >> 4;
  const int wave = threadIdx.x >> 6;
  const VOff o = v_offsets(d, W);
  const long P = (long)N * L;
  const long ntiles = (P + 15) / 16;
  // (Vh.b through the padding column W of the k-range, so that a layer's accumulators start from the inline constant zero
  //  instead of 12 bias registers read from LDS: built, same time -- LDS reads do not cost the SIMD's issue cycles)
  for (int idx = wave; idx < D::MTF * D::KS; idx += 4) {
    const int mt = idx / D::KS, ks = idx - mt * D::KS;
    sVh[idx * 64 + lane] = xw_fragA(ph + o.Vh, W, W, W, 16 * mt, 4 * ks);
  }
  if (threadIdx.x < 16 * D::MT) {
    sB[threadIdx.x] = (int)threadIdx.x < W ? ph[o.Vhb + threadIdx.x] : 0.0;
    sB[16 * D::MT + threadIdx.x] = (int)threadIdx.x < W ? ph[o.Vo + threadIdx.x] : 0.0;
    if (VINLDS || XPROJ) {
      sIn[threadIdx.x] = (int)threadIdx.x < W ? ph[o.Vin + (long)threadIdx.x * o.ldin] : 0.0;
      sIn[16 * D::MT + threadIdx.x] = (int)threadIdx.x < W ? ph[o.Vinb + threadIdx.x] : 0.0;
    }
  }
  const int ksd = (d + 3) / 4;
  const int ksl = ksd < VKS ? ksd : VKS;                       // k-steps served from LDS; ksl .. ksd - 1 from global memory
  if (VINLDS)
    for (int idx = wave; idx < ksl * D::MT; idx += 4) {
      const int ks = idx / D::MT, mt = idx - ks * D::MT;
      sVin[idx * 64 + lane] = xw_fragA(ph + o.Vin + 1, o.ldin, W, d, 16 * mt, 4 * ks);
    }
  if (D::VTAIL)
    for (int idx = threadIdx.x; idx < D::KS * 4 * D::TR; idx += blockDim.x) {
      const int r = idx % D::TR, k = idx / D::TR;                        // k = 4 ks + g
      sT[idx] = k < W ? ph[o.Vh + (long)(16 * (D::MT - 1) + r) * W + k] : 0.0;
    }
  __syncthreads();
  // (starting the two waves of a SIMD -- different blocks, launched together -- out of phase by part of a layer changes nothing:
  //  220.4 / 222.0 / 221.5 us at 0 / 2048 / 4096 clocks of offset, profiles/r06_disc_fwd_stagger.txt; a lone wave per SIMD already
  //  reaches 91 % of the two-wave rate)
  const double vob = ph[o.Vob];
  // Static split: wave gw takes tiles gw, gw + G, ... (G waves in the grid), so ntiles mod G waves carry one tile more
  // than the others.  The tiles of the first time index also run the fused reverse chain (about one more tile's worth of
  // work): the first round is rotated so that THEY land on waves with the smaller count -- the launch ends with the
  // slowest wave, 6 tile-times instead of 7 at the headline size (5.33 tiles per wave).
  const long G = (long)gridDim.x * 4;
  const long rot = ntiles >= G ? ntiles % G : 0;
  // (wave-uniform) sub-queues of this launch -- every one has a home block, which only stops once it is dry --, the one in
  // use, dry queues seen
  const int nq = (int)gridDim.x < XW_DISC_NQ ? (int)gridDim.x : XW_DISC_NQ;
  int cur = blockIdx.x % nq, tries = 0;
  int zr = 0;                                // zero, opaque to the compiler (see the layer's relu / gate)
  asm volatile("" : "+v"(zr));
#ifdef XW_CLOCK_PROBE
  const unsigned long long ck0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  for (long it = (long)blockIdx.x * 4 + wave; it < ntiles;) {
    const long tile = DYN ? it : (it < G ? it - rot + (it < rot ? G : 0) : it);
    long nxt = it + G;
    unsigned int ticket = 0;
    if (DYN && lane == 0) ticket = atomicAdd(queue + cur * XW_DISC_QSTRIDE, 1u);
    const Pt pt = locate(tile, P, N, tf, tpp);
    // nabla phi is only read at the first time index: the leading `ngrad` points (time-major
    // order) also get the input gradient of v, by a reverse chain through the masks stashed below
    const bool want_grad = gxv != nullptr && tile * 16 < ngrad;         // wave-uniform
    d4 a[D::MT], ad[D::MT];
    // The record is TILE-MAJOR: [tile of 16 points][(q+1) W rows][16] -- the 4 rows x 16 points a store instruction
    // covers are 512 contiguous bytes, and everything a wave writes (or k_disc_rec reads) for one tile is one 64 KB
    // stretch (row-major, the same accesses were 128-byte pieces 1 MB apart).  Every lane of the last tile has a slot.
    const long tu = __builtin_amdgcn_readfirstlane((int)tile);          // wave-uniform by construction: a scalar base
    double* actl = act + tu * ((long)(q + 1) * W * 16);                 // (laundered like pht below: the row addresses
    asm volatile("" : "+s"(actl));                                      //  are formed per tile, not hoisted)
    const int aoff = lane;                                              // (g, n) -> row offset g, point n
    const double* pht = ph;                                             // laundered: keeps the input-layer fragment
    asm volatile("" : "+s"(pht));                                       // addresses out of the loop-invariant (spilled) set
    if (XPROJ) {
      // input layer from the path's x-projection (rows >= W of it are zero): a_0 = (Vin[:, 1..d] x + Vin.b) + Vin[:, 0] t
      const double* xp = xproj + pt.n;
#pragma unroll
      for (int mt = 0; mt < D::MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          ad[mt][r] = sIn[16 * mt + g + 4 * r];
          a[mt][r] = (16 * mt + 4 * r < W) ? fma(ad[mt][r], pt.t, xw_ld_g(xp + (long)(16 * mt + 4 * r + g) * N)) : 0.0;
        }
    } else if (VINLDS) {
#pragma unroll
      for (int mt = 0; mt < D::MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          ad[mt][r] = sIn[16 * mt + g + 4 * r];
          a[mt][r] = sIn[16 * D::MT + 16 * mt + g + 4 * r] + ad[mt][r] * pt.t;
        }
      for (int ks = 0; ks < ksl; ++ks) {
        const int i = 4 * ks + g;
        const double b = i < d ? xT[(long)i * N + pt.n] : 0.0;
#pragma unroll
        for (int mt = 0; mt < D::MT; ++mt) a[mt] = XW_MFMA(sVin[(ks * D::MT + mt) * 64 + lane], b, a[mt]);
      }
      for (int ks = ksl; ks < ksd; ++ks) {                             // (d > 4 VKS: the rest of the input rows)
        const int i = 4 * ks + g;
        const double b = i < d ? xT[(long)i * N + pt.n] : 0.0;
#pragma unroll
        for (int mt = 0; mt < D::MT; ++mt) a[mt] = XW_MFMA(xw_fragA(pht + o.Vin + 1, o.ldin, W, d, 16 * mt, 4 * ks), b, a[mt]);
      }
    } else {
      input_layer<W>(pht, o, xT, N, d, pt, a, ad);
    }
    // relu'(+0) = 0 (torch): the layers below gate the d/dt tangent by the SIGN bit of the pre-activation, which is open at an
    // exact +0.  In a tied layer an exact +0 is a unit whose inputs are all dead -- its tangent is a sum of gated zeros -- but
    // HERE it can come from the input itself (t = 0, x = 0 with the zero bias the network starts from) with a tangent Vin[:, 0]
    // that is not zero: an exact zero leaves the input layer as -0, which every later test treats as "closed" (once per tile,
    // 2 vector instructions per register).
#pragma unroll
    for (int mt = 0; mt < D::MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double x = a[mt][r];
        a[mt][r] = __hiloint2double(x == 0.0 ? (int)0x80000000 : __double2hiint(x), __double2loint(x));
      }
    if (DYN) {
      nxt = G + cur + (long)nq * (unsigned int)__builtin_amdgcn_readfirstlane((int)ticket);
      while (nxt >= ntiles && tries < XW_DISC_STEAL && tries + 1 < nq) {      // home queue dry: the neighbours' (rare, end of the launch)
        ++tries;
        cur = cur + 1 < nq ? cur + 1 : 0;
        if (lane == 0) ticket = atomicAdd(queue + cur * XW_DISC_QSTRIDE, 1u);
        nxt = G + cur + (long)nq * (unsigned int)__builtin_amdgcn_readfirstlane((int)ticket);
      }
    }
    // one tied layer (value and d/dt tangent): (ai, adi) -> (nw, nd).  The loop below alternates two register sets so
    // that no layer ends with 32 register moves (a wave's VALU work does not overlap its FP64 MFMAs).
    auto layer = [&](int j, const d4 (&ai)[D::MT], const d4 (&adi)[D::MT], d4 (&nw)[D::MT], d4 (&nd)[D::MT]) {
#pragma unroll
      for (int mt = 0; mt < D::MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) nw[mt][r] = sB[16 * mt + g + 4 * r];
        nd[mt] = xw_zero4();
      }
      if (want_grad) {                                      // (wave-uniform; 1 tile in L needs the masks)
        unsigned int mask = 0;
#pragma unroll
        for (int mt = 0; mt < D::MT; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) mask |= (ai[mt][r] > 0.0 ? 1u : 0u) << (4 * mt + r);
        sMask[j][threadIdx.x] = mask;
      }
      double tv[D::TR], td[D::TR];                          // partial dot products of the tail rows (this lane's k = 4 ks + g)
#pragma unroll
      for (int r = 0; r < D::TR; ++r) tv[r] = td[r] = 0.0;
      // A-fragments (and the tail rows' weights) of k-step ks + 1 are requested from LDS BEFORE the matrix instructions of
      // k-step ks: a wave alone on its SIMD (640 of the 1024 SIMDs as launched beside the stepper) otherwise stops at a
      // wait in front of almost every pair of MFMAs
      double wn[D::MTF], tn[D::TR];
#pragma unroll
      for (int mt = 0; mt < D::MTF; ++mt) wn[mt] = sVh[(mt * D::KS) * 64 + lane];
      if (D::VTAIL)
#pragma unroll
        for (int r = 0; r < D::TR; ++r) tn[r] = sT[g * D::TR + r];
#pragma unroll
      for (int ks = 0; ks < D::KS; ++ks) {
        const double av = ai[ks >> 2][ks & 3];
        // relu and the tangent's gate from ONE 32-bit word, the sign extension of av's high word: x & ~sgn as a single
        // v_bfi_b32 per 32-bit half (sgn ? zr : x, zr a register that holds zero but is opaque to the compiler -- with a
        // visible zero, or a visible sign test, it rewrites the bit operations into compare + v_cndmask_b32 again).
        // 1 + 4 instructions of 2.3 clocks per k-step where v_max_f64 x 2 (canonicalise + max), v_or, v_mov, v_cmp_ne_u64 and
        // two v_cndmask_b32 took ~31.  av >= +0 keeps both, av < 0 (or -0) clears both: at av == +0 exactly the gate stays
        // open where torch's relu' is 0 -- in a tied layer that only happens where the tangent is zero as well (padding units,
        // dead inputs); the input layer's exact zeros arrive here as -0 (above).
        int sgn = __double2hiint(av) >> 31;
        asm volatile("" : "+v"(sgn));
        const double adv = adi[ks >> 2][ks & 3];
        const double b = __hiloint2double((sgn & zr) | (~sgn & __double2hiint(av)), (sgn & zr) | (~sgn & __double2loint(av)));
        const double bd = __hiloint2double((sgn & zr) | (~sgn & __double2hiint(adv)), (sgn & zr) | (~sgn & __double2loint(adv)));
        if (ACT) {   // activation store: row j W + k of the record is the layer input relu(a_j)[k], point-major
          double* __restrict__ rowp = actl + (j * W + 4 * ks) * 16;             // uniform pointer + 32-bit lane offset
          if (4 * ks + 3 < W || 4 * ks + g < W) xw_st_nt(b, rowp + aoff);   // (streamed: read once, much later)
        }
        double wc[D::MTF], tc[D::TR];
#pragma unroll
        for (int mt = 0; mt < D::MTF; ++mt) wc[mt] = wn[mt];
#pragma unroll
        for (int r = 0; r < D::TR; ++r) tc[r] = D::VTAIL ? tn[r] : 0.0;
        if (ks + 1 < D::KS) {
#pragma unroll
          for (int mt = 0; mt < D::MTF; ++mt) wn[mt] = sVh[(mt * D::KS + ks + 1) * 64 + lane];
          if (D::VTAIL)
#pragma unroll
            for (int r = 0; r < D::TR; ++r) tn[r] = sT[(4 * (ks + 1) + g) * D::TR + r];
        }
        asm volatile("" ::: "memory");                      // the requests stay HERE, one k-step ahead of their use
#pragma unroll
        for (int mt = 0; mt < D::MTF; ++mt) {
          nw[mt] = XW_MFMA(wc[mt], b, nw[mt]);
          nd[mt] = XW_MFMA(wc[mt], bd, nd[mt]);
        }
        if (D::VTAIL) {
#pragma unroll
          for (int r = 0; r < D::TR; ++r) {
            tv[r] = fma(tc[r], b, tv[r]);
            td[r] = fma(tc[r], bd, td[r]);
          }
        }
      }
      if (D::VTAIL && D::TR == 2) {
        // the four partials (value / tangent of rows 48, 49) -> ONE register with the totals in the four lane groups
        // (g = 0: row 48, g = 1: row 49, g = 2, 3: the two tangent totals), then the tangent's copy with the halves swapped.
        // Rows 50, 51 of both tiles (lane groups 2, 3) then hold finite leftovers instead of zeros: they only ever meet
        // zero-padded weights (columns >= W of the Vh fragments and of the tail rows, rows >= W of Vo and of Vh^T), and the
        // record's stores are guarded by row < W.  8 swaps + 4 adds where four separate sums took 16 + 8 and 8 selects.
        const double R = xw_fold16(xw_fold32(tv[0], td[0]), xw_fold32(tv[1], td[1]));
        d4 tw = xw_zero4(), tdd = xw_zero4();
        tw[0] = R + nw[D::MT - 1][0];
        const unsigned rl = __double2loint(R), rh = __double2hiint(R);
        typedef unsigned xw_u2 __attribute__((ext_vector_type(2)));
        const xw_u2 sl = __builtin_amdgcn_permlane32_swap(rl, rl, false, false), sh = __builtin_amdgcn_permlane32_swap(rh, rh, false, false);
        tdd[0] = __hiloint2double(sh[1], sl[1]);              // (R[32..63], R[32..63]): tangent totals in lane groups 0, 1
        nw[D::MT - 1] = tw;
        nd[D::MT - 1] = tdd;
      } else
      if (D::VTAIL) {   // row 16 (MT-1) + r of the chain layout lives in lane group g = r, register 0
        d4 tw = xw_zero4(), tdd = xw_zero4();
#pragma unroll
        for (int r = 0; r < D::TR; ++r) {
          const double sv_ = xw_sum_over_g(tv[r]) + nw[D::MT - 1][0], sd_ = xw_sum_over_g(td[r]);
          if (g == r) {
            tw[0] = sv_;
            tdd[0] = sd_;
          }
        }
        nw[D::MT - 1] = tw;
        nd[D::MT - 1] = tdd;
      }
    };
    {
      d4 b0[D::MT], bd0[D::MT];
      int j = 0;
      for (; j + 1 < q; j += 2) {
        layer(j, a, ad, b0, bd0);
        layer(j + 1, b0, bd0, a, ad);
      }
      if (j < q) {
        layer(j, a, ad, b0, bd0);
#pragma unroll
        for (int mt = 0; mt < D::MT; ++mt) {
          a[mt] = b0[mt];
          ad[mt] = bd0[mt];
        }
      }
    }
    double sv = 0.0, sd = 0.0;
#pragma unroll
    for (int mt = 0; mt < D::MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (16 * mt + 4 * r < W) {  // rows 16 mt + 4 r + g >= W only carry zero padding
          const double th = xw_tanh(a[mt][r]);
          if (ACT && (16 * mt + 4 * r + 3 < W || 16 * mt + 4 * r + g < W))
            xw_st_nt(th, actl + (q * W + 16 * mt + 4 * r) * 16 + aoff);   // last rows: tanh(a_q)
          const double vo = sB[16 * D::MT + 16 * mt + g + 4 * r];
          sv += vo * th;
          sd += vo * (1.0 - th * th) * ad[mt][r];
          a[mt][r] = vo * (1.0 - th * th);       // reused below as the cotangent of a_q for d(sum v)/d(input)
        } else {
          a[mt][r] = 0.0;
        }
    sv = xw_sum_over_g(sv) + vob;
    sd = xw_sum_over_g(sd);
    if (g == 0 && pt.valid) {
      v[pt.p] = sv;
      if (vt != nullptr) vt[pt.p] = sd;
    }
    if (want_grad) {
      // reverse chain: delta_j = relu'(a_j) .* (Vh^T delta_{j+1}); Vh^T fragments come from L2 (1 tile in L is affected).
      // The base pointer is laundered so that the per-lane fragment addresses are formed HERE: hoisted out of the tile
      // loop they spilled to scratch in every wave's prologue (17 MB of scratch writes per launch in the PMC pass).
      const double* phg = ph;
      asm volatile("" : "+s"(phg));
      int ll = lane;                         // (the lane-offset part of those addresses is loop-invariant too: a copy of the
      asm volatile("" : "+v"(ll));           //  lane id the compiler cannot see through keeps it here as well)
      d4 (&dl)[D::MT] = a;
      // (Gathering the A-fragments of Vh^T from the forward fragments in LDS -- element Vh[r][c] sits in fragment (r >> 4, c >> 2)
      //  at lane (r & 15) + 16 (c & 3), so a fragment of the transpose is one ds_read with an immediate offset plus a fixed lane
      //  part -- was built and is 8-way bank-conflicted: a lone gradient tile took 36 us more than a plain one, against 42 us
      //  with the loads inside the k-step and 29 us with the L2 loads one k-step ahead as below; tools/gradtile.py.)
      auto fragT = [&](int mt, int ks) -> double { return xw_fragAT_l(phg + o.Vh, W, W, W, 16 * mt, 4 * ks, ll); };
      for (int j = q - 1; j >= 0; --j) {
        d4 nd[D::MT];
#pragma unroll
        for (int mt = 0; mt < D::MT; ++mt) nd[mt] = xw_zero4();
        // (the fragments of k-step ks + 1 are requested before the matrix instructions of k-step ks)
        double fn[D::MT];
#pragma unroll
        for (int mt = 0; mt < D::MT; ++mt) fn[mt] = fragT(mt, 0);
#pragma unroll
        for (int ks = 0; ks < D::KS; ++ks) {
          const double b = dl[ks >> 2][ks & 3];
          double fc[D::MT];
#pragma unroll
          for (int mt = 0; mt < D::MT; ++mt) fc[mt] = fn[mt];
          if (ks + 1 < D::KS) {
#pragma unroll
            for (int mt = 0; mt < D::MT; ++mt) fn[mt] = fragT(mt, ks + 1);
          }
          asm volatile("" ::: "memory");   // a few loads in flight, not all 52 (register pressure)
#pragma unroll
          for (int mt = 0; mt < D::MT; ++mt) nd[mt] = XW_MFMA(fc[mt], b, nd[mt]);
        }
        const unsigned int mask = sMask[j][threadIdx.x];
#pragma unroll
        for (int mt = 0; mt < D::MT; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) dl[mt][r] = ((mask >> (4 * mt + r)) & 1u) ? nd[mt][r] : 0.0;
      }
      int ngl = ngrad;                       // laundered: keeps the (lane-dependent) row offsets i * ngrad of the gxv stores
      asm volatile("" : "+s"(ngl));          // out of the loop-invariant set (they were spilled in every wave's prologue)
      const bool gl = pt.valid && pt.p < ngl;
      for (int rt = 0; rt < (d + 15) / 16; ++rt) {
        d4 vv = xw_zero4();
#pragma unroll
        for (int ks = 0; ks < D::KS; ++ks) {
          if ((ks & 3) == 0) asm volatile("" ::: "memory");
          vv = XW_MFMA(xw_fragAT_l(phg + o.Vin + 1, o.ldin, W, d, 16 * rt, 4 * ks, ll), dl[ks >> 2][ks & 3], vv);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * rt + g + 4 * r;
          if (i < d && gl) gxv[(long)i * ngl + pt.p] = vv[r];
        }
      }
      double st_ = 0.0;
#pragma unroll
      for (int mt = 0; mt < D::MT; ++mt) {
        const d4 v0 = xw_vecD_strided(phg + o.Vin, o.ldin, W, 16 * mt);
#pragma unroll
        for (int r = 0; r < 4; ++r) st_ += v0[r] * dl[mt][r];
      }
      st_ = xw_sum_over_g(st_);
      if (gtv != nullptr && g == 0 && gl) gtv[pt.p] = st_;
    }
    it = nxt;
  }
#ifdef XW_CLOCK_PROBE
  if (lane == 0 && blockIdx.x * 4 + wave < 4096) {
    xw_clock_buf[2 * (blockIdx.x * 4 + wave)] = __builtin_amdgcn_s_memtime() - ck0;
    xw_clock_buf[2 * (blockIdx.x * 4 + wave) + 1] = __builtin_amdgcn_s_memrealtime() - rt0;
  }
#endif
  if (DYN && lane == 0) {
    if (atomicAdd(&sDone, 1u) == 3u && atomicAdd(queue + XW_DISC_NQ * XW_DISC_QSTRIDE, 1u) == gridDim.x - 1u)
      for (int s_ = 0; s_ <= XW_DISC_NQ; ++s_) atomicExch(queue + s_ * XW_DISC_QSTRIDE, 0u);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// LDS plan of k_disc_bwd (doubles):  Vh frags | Vh^T frags | delta T-tiles [4 waves][MT] | r T-tiles [4][MT] | dVo acc
template <int W> struct BwdLds {
  typedef VDim<W> D;
  static constexpr int nfrag = D::MT * D::KS * 64;
  static constexpr int ntt = 4 * D::MT * XW_TTILE;
  static constexpr int oVh = 0, oVhT = nfrag, oD = 2 * nfrag, oR = 2 * nfrag + ntt, oO = 2 * nfrag + 2 * ntt;
  static constexpr int oB = oO + 4 * 64 * 16;     // Vh.b and Vo, zero-padded to 16 MT rows each
  static constexpr int total = oB + 2 * 16 * D::MT;
  static_assert(total * 8 <= 160 * 1024, "LDS budget of one CU");
};

// (the parameter gradient from the activation record is k_disc_rec below; this kernel recomputes the forward per tile)
template <int W, int Q, int CTG, bool PARAMS, bool INGRAD>
__global__ void __launch_bounds__(256) k_disc_bwd(const double* __restrict__ xT, const double* __restrict__ tf,
                                                  const double* __restrict__ tpp, const double* __restrict__ ph,
                                                  const double* __restrict__ vbar, int N, int L, int d,
                                                  double* __restrict__ gslab, double* __restrict__ gxv,
                                                  double* __restrict__ gtv) {
  typedef VDim<W> D;
  typedef BwdLds<W> S;
  static_assert(D::MT == 4, "the recomputing reverse kernel maps the 4 row tiles of dVh onto the 4 waves of a block");
  static_assert(D::BIASROW, "the recomputing reverse kernel takes dVh.b from the ones row");
  __shared__ double lds[S::total];
  double* sVh = lds + S::oVh;
  double* sVhT = lds + S::oVhT;
  double* sD = lds + S::oD;
  double* sR = lds + S::oR;
  double* sO = lds + S::oO;
  const int lane = xw_lane(), g = lane >> 4, n = lane & 15;
  const int wave = threadIdx.x >> 6;
  const VOff o = v_offsets(d, W);
  const long P = (long)N * L;
  const long nsuper = (P + 63) / 64;

  double* sB = lds + S::oB;
  for (int idx = wave; idx < D::MT * D::KS; idx += 4) {
    const int mt = idx / D::KS, ks = idx - mt * D::KS;
    sVh[idx * 64 + lane] = xw_fragA(ph + o.Vh, W, W, W, 16 * mt, 4 * ks);
    sVhT[idx * 64 + lane] = xw_fragAT(ph + o.Vh, W, W, W, 16 * mt, 4 * ks);
  }
  if (threadIdx.x < 16 * D::MT) {
    sB[threadIdx.x] = (int)threadIdx.x < W ? ph[o.Vhb + threadIdx.x] : 0.0;
    sB[16 * D::MT + threadIdx.x] = (int)threadIdx.x < W ? ph[o.Vo + threadIdx.x] : 0.0;
  }
  if (PARAMS)
    for (int i = 0; i < 16; ++i) sO[i * 256 + wave * 64 + lane] = 0.0;   // [slot][thread]: lane-contiguous, no bank conflicts
  __syncthreads();

  d4 accH[D::MT], accIn[CTG * 4];
#pragma unroll
  for (int ct = 0; ct < D::MT; ++ct) accH[ct] = xw_zero4();
#pragma unroll
  for (int ct = 0; ct < CTG * 4; ++ct) accIn[ct] = xw_zero4();

  // one tied hidden layer  a_out = Vh.b + Vh r  with A-fragments and the bias read from LDS
  auto layer = [&](const d4 (&r)[D::MT], d4 (&out)[D::MT]) {
    asm volatile("" ::: "memory");   // keep the A-fragments in LDS: without this the compiler hoists all 52 loads of
                                     // every unrolled layer into registers and spills the activations instead
#pragma unroll
    for (int mt = 0; mt < D::MT; ++mt)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) out[mt][rr] = sB[16 * mt + g + 4 * rr];
#pragma unroll
    for (int ks = 0; ks < D::KS; ++ks) {
      const double b = r[ks >> 2][ks & 3];
#pragma unroll
      for (int mt = 0; mt < D::MT; ++mt) out[mt] = XW_MFMA(sVh[(mt * D::KS + ks) * 64 + lane], b, out[mt]);
    }
  };

  // layers per checkpoint segment.  1: every layer input r_j stays in registers (13 live f64 per layer at W = 50) and
  // nothing is recomputed -- 497 vs 558 us at SEG = 3 even with 48 spilled registers.  The wide-input variant (CTG = 2,
  // d > 62) carries 32 more accumulator registers and keeps the 3-layer segments.
  constexpr int SEG = CTG == 1 ? 1 : 3;
  constexpr int NSEG = (Q + SEG - 1) / SEG;
  for (long st = blockIdx.x; st < nsuper; st += gridDim.x) {
    const Pt pt = locate(st * 4 + wave, P, N, tf, tpp);
    // ---- forward: only r_0, r_3, r_6 (= relu(a_j)) are kept; the layers in between are recomputed per segment, so the
    //      live activations are 2 x SEG tiles instead of Q (which did not fit the 512-register file and spilled)
    d4 ck[NSEG][D::MT];
    d4 a[D::MT], ad[D::MT];
    input_layer<W>(ph, o, xT, N, d, pt, a, ad);
#pragma unroll
    for (int j = 0; j < Q; ++j) {
      d4 r[D::MT];
#pragma unroll
      for (int mt = 0; mt < D::MT; ++mt) {
#pragma unroll
        for (int q_ = 0; q_ < 4; ++q_)
          if (q_ < D::LR(mt)) {
            r[mt][q_] = a[mt][q_] > 0.0 ? a[mt][q_] : 0.0;
            if (j % SEG == 0) ck[j / SEG][mt][q_] = r[mt][q_];
          }
      }
      layer(r, a);
    }
    // ---- output layer and its cotangent
    const double vb = pt.valid ? (vbar != nullptr ? vbar[pt.p] : 1.0) : 0.0;
    d4 dl[D::MT];
#pragma unroll
    for (int mt = 0; mt < D::MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double th = 0.0;
        if (16 * mt + 4 * r < W) th = xw_tanh(a[mt][r]);
        dl[mt][r] = sB[16 * D::MT + 16 * mt + g + 4 * r] * (1.0 - th * th) * vb;
        if (PARAMS) sO[(mt * 4 + r) * 256 + wave * 64 + lane] += (mt == D::MT - 1 && r == 3) ? (g == 0 ? vb : 0.0) : vb * th;
      }
    // ---- reverse chain, segment by segment
#pragma unroll
    for (int sg = NSEG - 1; sg >= 0; --sg) {
      d4 seg[SEG][D::MT];
#pragma unroll
      for (int mt = 0; mt < D::MT; ++mt)
#pragma unroll
        for (int q_ = 0; q_ < 4; ++q_)
          if (q_ < D::LR(mt)) seg[0][mt][q_] = ck[sg][mt][q_];
#pragma unroll
      for (int k = 1; k < SEG; ++k)
        if (sg * SEG + k < Q) {
          d4 tmp[D::MT];
          layer(seg[k - 1], tmp);
#pragma unroll
          for (int mt = 0; mt < D::MT; ++mt)
#pragma unroll
            for (int q_ = 0; q_ < 4; ++q_)
              if (q_ < D::LR(mt)) seg[k][mt][q_] = tmp[mt][q_] > 0.0 ? tmp[mt][q_] : 0.0;
        }
#pragma unroll
      for (int k = SEG - 1; k >= 0; --k) {
        if (sg * SEG + k >= Q) continue;
        if (PARAMS) {
#pragma unroll
          for (int mt = 0; mt < D::MT; ++mt) {
            d4 rj = seg[k][mt];
            if (mt == (W >> 4)) {
              if (g == ((W & 15) & 3)) rj[(W & 15) >> 2] = 1.0;  // ones row -> column W of dVh collects dVh.b
            }
            if (mt < D::MT - 1) {
              xw_writeT(sD + (wave * D::MT + mt) * XW_TTILE, dl[mt]);
              xw_writeT(sR + (wave * D::MT + mt) * XW_TTILE, rj);
            } else {
              xw_writeT_n<D::LR(D::MT - 1)>(sD + (wave * D::MT + mt) * XW_TTILE, dl[mt]);
              xw_writeT_n<D::LR(D::MT - 1)>(sR + (wave * D::MT + mt) * XW_TTILE, rj);
            }
          }
        }
        // reverse chain first: its 52 MFMAs only need dl and Vh^T and cover the latency of the LDS transposes above
        d4 nd[D::MT];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int mt = 0; mt < D::MT; ++mt) nd[mt] = xw_zero4();
#pragma unroll
        for (int ks = 0; ks < D::KS; ++ks) {
          const double b = dl[ks >> 2][ks & 3];
#pragma unroll
          for (int mt = 0; mt < D::MT; ++mt) nd[mt] = XW_MFMA(sVhT[(mt * D::KS + ks) * 64 + lane], b, nd[mt]);
        }
        if (PARAMS) {
          __syncthreads();
#pragma unroll
          for (int pw = 0; pw < 4; ++pw)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
              const double av = xw_readT(sD + (pw * D::MT + wave) * XW_TTILE, ks);
#pragma unroll
              for (int ct = 0; ct < D::MT; ++ct)
                accH[ct] = XW_MFMA(av, xw_readT(sR + (pw * D::MT + ct) * XW_TTILE, ks), accH[ct]);
            }
          __syncthreads();
        }
#pragma unroll
        for (int mt = 0; mt < D::MT; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (r < D::LR(mt)) dl[mt][r] = seg[k][mt][r] > 0.0 ? nd[mt][r] : 0.0;
      }
    }
    // ---- dl = cotangent of a_0.  Input layer: dVin = dl (x) [t; x; 1]
    if (PARAMS) {
#pragma unroll
      for (int mt = 0; mt < D::MT; ++mt) xw_writeT(sD + (wave * D::MT + mt) * XW_TTILE, dl[mt]);
#pragma unroll
      for (int grp = 0; grp < CTG; ++grp) {
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
          const int cl = g + 4 * rr;            // local row 0..63 of this group
          const int c = 64 * grp + cl;          // input row: 0 = t, 1..d = x, d+1 = ones
          double val = 0.0;
          if (pt.valid) {
            if (c == 0) val = pt.t;
            else if (c <= d) val = xT[(long)(c - 1) * N + pt.n];
            else if (c == d + 1) val = 1.0;
          }
          sR[(wave * D::MT + (cl >> 4)) * XW_TTILE + (cl & 15) * XW_TSTRIDE + n] = val;
        }
        __syncthreads();
#pragma unroll
        for (int pw = 0; pw < 4; ++pw)
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            const double av = xw_readT(sD + (pw * D::MT + wave) * XW_TTILE, ks);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
              accIn[grp * 4 + ct] = XW_MFMA(av, xw_readT(sR + (pw * D::MT + ct) * XW_TTILE, ks), accIn[grp * 4 + ct]);
          }
        __syncthreads();
      }
    }
    if (INGRAD) {
      for (int rt = 0; rt < (d + 15) / 16; ++rt) {
        d4 vv = xw_zero4();
#pragma unroll
        for (int ks = 0; ks < D::KS; ++ks)
          vv = XW_MFMA(xw_fragAT(ph + o.Vin + 1, o.ldin, W, d, 16 * rt, 4 * ks), dl[ks >> 2][ks & 3], vv);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * rt + g + 4 * r;
          if (i < d && pt.valid) gxv[(long)i * P + pt.p] = vv[r];
        }
      }
      double st_ = 0.0;
#pragma unroll
      for (int mt = 0; mt < D::MT; ++mt) {
        const d4 v0 = xw_vecD_strided(ph + o.Vin, o.ldin, W, 16 * mt);
#pragma unroll
        for (int r = 0; r < 4; ++r) st_ += v0[r] * dl[mt][r];
      }
      st_ = xw_sum_over_g(st_);
      if (gtv != nullptr && g == 0 && pt.valid) gtv[pt.p] = st_;
    }
  }

  if (PARAMS) {
    double* slab = gslab + (long)blockIdx.x * o.total;
    // wave `wave` owns rows [16 wave, 16 wave + 16) of dVh and dVin
#pragma unroll
    for (int ct = 0; ct < D::MT; ++ct) {
      const int c = 16 * ct + n;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * wave + g + 4 * r;
        if (row < W) {
          if (c < W) slab[o.Vh + row * W + c] = accH[ct][r];
          else if (c == W) slab[o.Vhb + row] = accH[ct][r];
        }
      }
    }
#pragma unroll
    for (int ct = 0; ct < CTG * 4; ++ct) {
      const int c = 16 * ct + n;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * wave + g + 4 * r;
        if (row < W) {
          if (c <= d) slab[o.Vin + row * o.ldin + c] = accIn[ct][r];
          else if (c == d + 1) slab[o.Vinb + row] = accIn[ct][r];
        }
      }
    }
    __syncthreads();
    const int tid = threadIdx.x;
    if (tid <= W) {
      double s = 0.0;
      if (tid < W) {
        const int mt = tid >> 4, gg = tid & 3, r = (tid & 15) >> 2;
        for (int wv = 0; wv < 4; ++wv)
          for (int nn = 0; nn < 16; ++nn) s += sO[(mt * 4 + r) * 256 + wv * 64 + gg * 16 + nn];
        slab[o.Vo + tid] = s;
      } else {
        for (int wv = 0; wv < 4; ++wv)
          for (int nn = 0; nn < 16; ++nn) s += sO[15 * 256 + wv * 64 + nn];
        slab[o.Vob] = s;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// k_disc_rec: the parameter gradient from the activation record, two blocks per CU.
// k_disc_bwd above needs 157 KB of LDS (one block per CU, one wave per SIMD: every barrier / transpose stall of that
// wave is idle matrix-pipe time).  This kernel only ever runs from the record, so
//   * the forward fragments of Vh are gone, and of Vh^T only the three full row tiles stay in LDS: rows 48, 49 of the
//     reverse chain (W = 50) are contracted on the vector ALU from a 104-double table, as in k_disc_fwd -- 39 instead of
//     52 chain MFMAs per layer;
//   * the last transposed tile of every set stores its one live 4-row group (rows 48..51) instead of 16 rows; operand
//     rows read past it (lane & 15 >= 4, folded back with & 3) only reach accumulator rows / columns that are never stored;
//   * the dVo partial sums are reduced over the 16 points of a tile before they go to LDS (256 instead of 4096 doubles);
//   * the input-layer gradient is contracted in groups of 48 input rows (three full tiles);
// -> 78 KB per block, <= 256 registers per wave: two blocks share a CU and cover each other's barriers.
#ifdef XW_REC_PROBE     // diagnostic build only (tools/probe_rec_phases.py): shader clocks per phase of every wave of k_disc_rec
__device__ unsigned long long xw_rec_clock[12 * 2048];
#define XW_PH(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); phc[i] += t_ - phl; phl = t_; }
#else
#define XW_PH(i)
#endif
template <int W> struct RecLds {
  typedef VDim<W> D;
  // W = 49..51: short last row tile -- vector-ALU tail + one live 4-row group, 78 KB, two blocks per CU.  Otherwise
  // (W = 64, the container of the widths above 50) four full tiles everywhere: 105 KB, one block per CU.
  static constexpr bool VT = D::VTAIL;
  static_assert(!VT || D::LR(D::MT - 1) == 1, "short last row tile: vector-ALU tail + one live 4-row group");
  static constexpr int T3 = VT ? 4 * XW_TSTRIDE : XW_TTILE;       // trimmed last tile
  static constexpr int wset = (D::MT - 1) * XW_TTILE + T3;        // one wave's set of transposed tiles
  // W = 128: the 131 KB of Vh^T fragments do not fit beside the tile sets (139 KB) -- the reverse chain takes them from L2,
  // one k-step ahead, like the fused input gradient of k_disc_fwd
  static constexpr bool FRAG_LDS = D::MT == 4;
  static constexpr int oVhT = 0;                                  // [MTF][KS][64]
  static constexpr int oTT = oVhT + (FRAG_LDS ? D::MTF * D::KS * 64 : 0);   // [4 KS][TR]: Vh[k][16 (MT-1) + r]
  static constexpr int oVo = oTT + (VT ? 4 * D::KS * D::TR : 0);  // Vo, zero-padded to 16 MT rows
  static constexpr int oD = oVo + 16 * D::MT;
  static constexpr int oR = oD + 4 * wset;
  static constexpr int oO = oR + 4 * wset;                        // [4 MT row slots + dVo.b][4 waves][4 g]
  static constexpr int NO = 4 * D::MT;                            // row slots of dVo
  static constexpr int total = oO + (NO + 1) * 16;
  static_assert(total * 8 <= (VT ? 80 : 160) * 1024, "two blocks per CU (short tail) / one");
  __device__ static constexpr int toff(int mt) { return mt * XW_TTILE; }
};

// operand read from a wave's tile set: tile mt, folded back into the live rows when it is the trimmed one
template <int W> __device__ __forceinline__ double rec_readT(const double* set, int mt, int ks) {
  const int l = xw_lane();
  const int row = (RecLds<W>::VT && mt == VDim<W>::MT - 1) ? (l & 3) : (l & 15);
  return set[RecLds<W>::toff(mt) + row * XW_TSTRIDE + 4 * ks + (l >> 4)];
}

// TSUM (path mode, N a multiple of 64: every 64-point unit is 64 consecutive paths at ONE time index): the input layer's
//   gradient dVin = sum_p delta_0[p] (x) [t_p ; x_p ; 1] is not contracted unit by unit.  A block takes a CONTIGUOUS run of units in
//   (path group, time index) order -- its waves stay on the same 16 paths while the time index advances -- and x does not move
//   along a vertical path, so  dVin[:, 1..d] = (sum_l delta_0) (x) x,  dVin.b = rowsum(sum_l delta_0),  dVin[:, 0] = rowsum(sum_l
//   t_l delta_0):  two running sums in registers per unit (26 multiply-adds) and ONE contraction per path group instead of one
//   per unit (per unit it was 12 gathered loads, 12 LDS stores, two barriers, 32 matrix instructions and a read-modify-write of
//   the slab: 11 % of a wave's time, profiles/r04_probe_rec_phases.txt).
template <int W, int Q, int NG, bool TSUM>
__global__ void __launch_bounds__(256, RecLds<W>::VT ? 2 : 1) k_disc_rec(const double* __restrict__ xT, const double* __restrict__ tf,
                                                     const double* __restrict__ tpp, const double* __restrict__ ph,
                                                     const double* __restrict__ vbar, int N, int L, int d,
                                                     double* __restrict__ gslab, const double* __restrict__ act, int qrt) {
  typedef VDim<W> D;
  typedef RecLds<W> S;
  __shared__ double lds[S::total];
  double* sVhT = lds + S::oVhT;
  double* sTT = lds + S::oTT;
  double* sVo = lds + S::oVo;
  double* sO = lds + S::oO;
  const int lane = xw_lane(), g = lane >> 4, n = lane & 15;
  const int wave = threadIdx.x >> 6;
  double* myD = lds + S::oD + wave * S::wset;
  double* myR = lds + S::oR + wave * S::wset;
#ifdef XW_REC_PROBE
  const unsigned long long ck_in = __builtin_amdgcn_s_memtime(), rt_in = __builtin_amdgcn_s_memrealtime();
#endif
  const VOff o = v_offsets(d, W);
  const long P = (long)N * L;
  const long nsuper = (P + 63) / 64;
  const int nq = Q > 0 ? Q : qrt;
  constexpr int UNROLL = Q > 0 ? Q : 1;
  constexpr int TW = D::MT / 2;                      // 2 x 2 waves x TW x TW tiles of dVh
  const int wi = wave >> 1, wj = wave & 1;

  if (S::FRAG_LDS)
    for (int idx = wave; idx < D::MTF * D::KS; idx += 4) {
      const int mt = idx / D::KS, ks = idx - mt * D::KS;
      sVhT[idx * 64 + lane] = xw_fragAT(ph + o.Vh, W, W, W, 16 * mt, 4 * ks);
    }
  if (S::VT)
    for (int idx = threadIdx.x; idx < 4 * D::KS * D::TR; idx += blockDim.x) {
      const int r = idx % D::TR, k = idx / D::TR;                        // reverse chain: nd[48 + r] = sum_k Vh[k][48 + r] dl[k]
      sTT[idx] = k < W ? ph[o.Vh + (long)k * W + 16 * (D::MT - 1) + r] : 0.0;
    }
  if (threadIdx.x < 16 * D::MT) sVo[threadIdx.x] = (int)threadIdx.x < W ? ph[o.Vo + threadIdx.x] : 0.0;
  for (int idx = threadIdx.x; idx < (S::NO + 1) * 16; idx += blockDim.x) sO[idx] = 0.0;
  __syncthreads();

  d4 accH[TW * TW], accB[D::MT];                     // accB: dVh.b without the ones row (W a multiple of 16)
#pragma unroll
  for (int ct = 0; ct < TW * TW; ++ct) accH[ct] = xw_zero4();
#pragma unroll
  for (int ct = 0; ct < D::MT; ++ct) accB[ct] = xw_zero4();
  // ---- short last row tile (W = 50): dVh = 48 x 48 core on 16x16x4 tiles + the two edges on 4x4x4 blocks.
  // As 4 x 4 tiles of 16 x 16 the outer products ran 16 matrix instructions per k-step for 50 x 51 live entries of 64 x 64
  // (61 %): 7 of the 16 tiles existed for rows 48, 49 / columns 48..50.  Now, per 16 points,
  //     core  : 9 tiles x 4 k-steps of 66 clocks -- two tiles per wave (accH[0], accH[1]) and tile (2, 2) of the wave's OWN
  //             16 points (accH[2]: a partial sum per wave, needs no barrier; the four are added once, at the end);
  //     edges : rows 48..51 x column blocks 0..12 (waves 0, 1) and row blocks 0..11 x columns 48..51 (waves 2, 3) as
  //             v_mfma_f64_4x4x4 -- the instruction's four blocks are the four groups of four points, so ONE instruction
  //             is a 4 x 4 block of dVh over all 16 points (18 clocks; the four partials are folded once, at the end).
  //             The column edge is computed transposed (A = input rows 48..51, B = cotangent rows), so that all four waves
  //             run the same code: one fixed operand X, 6-7 varying operands V.
  // 2826 instead of 4224 matrix clocks per 16 points and layer, 706 +- 14 per wave.
  constexpr bool EDGE = S::VT;
  constexpr int NE = 7;
  double accE[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) accE[e] = 0.0;
  // core tiles: waves 0..2 row band `wave` x column tiles 0, 1 (one cotangent operand, two input operands per k-step); wave 3
  // column tile 2 x row bands 0, 1, computed transposed (A = input tile 2, B = the two cotangent tiles): three operand
  // reads per two matrix instructions in every wave, the same code in all of them
  const int vstart = wave == 1 ? 7 : wave == 3 ? 6 : 0;
  const int lane16 = (lane & 15) * XW_TSTRIDE + (lane >> 4);                       // 16x16x4 operand: row i, point 4 ks + k
  const int lane4 = (lane & 3) * XW_TSTRIDE + ((lane >> 2) & 3) * 4 + (lane >> 4);  // 4x4x4 operand: row i, point 4 blk + k
  const int offC = (wave < 3 ? S::oD + wave * XW_TTILE : S::oR + 2 * XW_TTILE) + lane16;
  const int offW = (wave < 3 ? S::oR : S::oD) + lane16;
  const int offX = (wave < 2 ? S::oD : S::oR) + 48 * XW_TSTRIDE + lane4;
  const int offV = (wave < 2 ? S::oR : S::oD) + vstart * 4 * XW_TSTRIDE + lane4;
  double* slab = gslab + (long)blockIdx.x * o.total;

#ifdef XW_REC_PROBE
  unsigned long long phc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, phl = __builtin_amdgcn_s_memtime();
#endif
  // TSUM: this block's units [u0, u1) in (path group, time index) order; unit u = (ng, l) is super-tile l (N / 64) + ng
  const long ngs = N >> 6;
  const long ub = nsuper / gridDim.x, ur = nsuper % gridDim.x;
  const long u0 = (long)blockIdx.x * ub + ((long)blockIdx.x < ur ? (long)blockIdx.x : ur), u1 = u0 + ub + ((long)blockIdx.x < ur ? 1 : 0);
  d4 sumS[D::MT], sumT[D::MT];                       // TSUM: sum_l delta_0, sum_l t_l delta_0 of the current path group
#pragma unroll
  for (int mt = 0; mt < D::MT; ++mt) sumS[mt] = sumT[mt] = xw_zero4();
  bool slab_first = true;                            // (uniform) the block's slab has not been written yet
  for (long it = TSUM ? u0 : (long)blockIdx.x; it < (TSUM ? u1 : nsuper); it += TSUM ? 1 : (long)gridDim.x) {
    const long ung = TSUM ? it / L : 0;
    const long st = TSUM ? (it - ung * L) * ngs + ung : it;
    const Pt pt = locate(st * 4 + wave, P, N, tf, tpp);
    // tile-major record (see k_disc_fwd): this wave's tile is one contiguous stretch, a layer 13 x 512 contiguous bytes
    const long ntile = (P + 15) >> 4;
    const long tl = st * 4 + wave < ntile ? st * 4 + wave : ntile - 1;  // (a wave past the end re-reads the last tile; vb = 0)
    const double* tbase = act + (long)__builtin_amdgcn_readfirstlane((int)tl) * ((long)(nq + 1) * W * 16);
    const int aoff = lane;
    auto load_layer = [&](int j, d4 (&r)[D::MT]) {
      const double* base = tbase + j * W * 16;
      asm volatile("" : "+s"(base));
#pragma unroll
      for (int mt = 0; mt < D::MT; ++mt)
#pragma unroll
        for (int q_ = 0; q_ < 4; ++q_)
          if (q_ < D::LR(mt)) {
            const int row = 16 * mt + 4 * q_ + g;
            r[mt][q_] = (16 * mt + 4 * q_ + 3 < W || row < W) ? xw_ld_nt(base + (16 * mt + 4 * q_) * 16 + aoff) : 0.0;
          }
    };
    // (requesting the first two layers of the NEXT tile in front of this tile's input-layer gradient, so that a tile does not
    //  start with a round trip to memory: 13 spilled registers, no gain -- the other wave of the SIMD covers that wait)
    d4 a[D::MT], rnext[D::MT];
    load_layer(nq, a);                                 // tanh(a_q)
    if (nq > 0) load_layer(nq - 1, rnext);
    const double vb = pt.valid ? (vbar != nullptr ? xw_ld_g(vbar + pt.p) : 1.0) : 0.0;
    // ---- output layer: cotangent of a_q, and dVo / dVo.b reduced over the 16 points of the tile
    d4 dl[D::MT];
#pragma unroll
    for (int mt = 0; mt < D::MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (r < D::LR(mt) && 16 * mt + 4 * r < W) {
          const double th = a[mt][r];
          dl[mt][r] = sVo[16 * mt + g + 4 * r] * (1.0 - th * th) * vb;
          const double s_ = xw_sum_over_n(vb * th);
          if (n == 0) sO[(mt * 4 + r) * 16 + wave * 4 + g] += s_;
        } else {
          dl[mt][r] = 0.0;
        }
      }
    {
      const double s_ = xw_sum_over_n(vb);
      if (lane == 0) sO[S::NO * 16 + wave * 4] += s_;
    }
    XW_PH(6)
    // ---- reverse chain, one layer at a time; the next layer's inputs are in flight meanwhile
#pragma unroll UNROLL
    for (int sg = nq - 1; sg >= 0; --sg) {
      // the layer's inputs are only needed transposed in LDS and, afterwards, as the ReLU mask: 13 sign bits, not 26 registers
      bool open[4 * D::MT];                           // (lane masks in scalar register pairs: one compare now, one select later)
#pragma unroll
      for (int mt = 0; mt < D::MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) open[4 * mt + r] = r < D::LR(mt) && rnext[mt][r] > 0.0;
#pragma unroll
      for (int mt = 0; mt < D::MT; ++mt) {
        d4 rr = rnext[mt];
        if (D::BIASROW && mt == (W >> 4)) {
          if (g == ((W & 15) & 3)) rr[(W & 15) >> 2] = 1.0;            // ones row -> column W of dVh collects dVh.b
        }
        if (!D::BIASROW) accB[mt] = accB[mt] + dl[mt];
        if (mt < D::MT - 1 || !S::VT) {
          xw_writeT(myD + S::toff(mt), dl[mt]);
          xw_writeT(myR + S::toff(mt), rr);
        } else {
          xw_writeT_n<1>(myD + S::toff(mt), dl[mt]);
          xw_writeT_n<1>(myR + S::toff(mt), rr);
        }
      }
      if (sg > 0) load_layer(sg - 1, rnext);                           // in flight while this layer is reversed
      XW_PH(0)
      // reverse chain: three row tiles on the matrix pipe, rows 16 (MT-1) + r on the vector ALU
      d4 nd[D::MT];
      asm volatile("" ::: "memory");
#pragma unroll
      for (int mt = 0; mt < D::MT; ++mt) nd[mt] = xw_zero4();
      double tv[D::TR];
#pragma unroll
      for (int r = 0; r < D::TR; ++r) tv[r] = 0.0;
      if constexpr (S::FRAG_LDS) {
#pragma unroll
        for (int ks = 0; ks < D::KS; ++ks) {
          const double b = dl[ks >> 2][ks & 3];
          if ((ks & 1) == 0) asm volatile("" ::: "memory");   // bound the LDS loads in flight (else they are all hoisted -> spills)
#pragma unroll
          for (int mt = 0; mt < D::MTF; ++mt) nd[mt] = XW_MFMA(sVhT[(mt * D::KS + ks) * 64 + lane], b, nd[mt]);
          if (S::VT) {
#pragma unroll
            for (int r = 0; r < D::TR; ++r) tv[r] = fma(sTT[(4 * ks + g) * D::TR + r], b, tv[r]);
          }
        }
      } else {
        // the wide container: Vh^T fragments from L2, requested one k-step ahead (laundered base and lane id: the per-lane
        // fragment addresses are formed here, not hoisted out of the tile loop into spilled registers)
        const double* phg = ph;
        int ll = lane;
        asm volatile("" : "+s"(phg), "+v"(ll));
        double fn[D::MT];
#pragma unroll
        for (int mt = 0; mt < D::MT; ++mt) fn[mt] = xw_fragAT_l(phg + o.Vh, W, W, W, 16 * mt, 0, ll);
#pragma unroll
        for (int ks = 0; ks < D::KS; ++ks) {
          const double b = dl[ks >> 2][ks & 3];
          double fc[D::MT];
#pragma unroll
          for (int mt = 0; mt < D::MT; ++mt) fc[mt] = fn[mt];
          if (ks + 1 < D::KS) {
#pragma unroll
            for (int mt = 0; mt < D::MT; ++mt) fn[mt] = xw_fragAT_l(phg + o.Vh, W, W, W, 16 * mt, 4 * (ks + 1), ll);
          }
          asm volatile("" ::: "memory");
#pragma unroll
          for (int mt = 0; mt < D::MT; ++mt) nd[mt] = XW_MFMA(fc[mt], b, nd[mt]);
        }
      }
      if (S::VT) {
#pragma unroll
        for (int r = 0; r < D::TR; ++r) {
          const double s_ = xw_sum_over_g(tv[r]);
          if (g == r) nd[D::MT - 1][0] = s_;
        }
      }
      XW_PH(1)
      if (EDGE) {
        // tile (2, 2) over the wave's own points: its own transposed tiles, no barrier needed
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          accH[2] = XW_MFMA(xw_readT(myD + S::toff(2), ks), xw_readT(myR + S::toff(2), ks), accH[2]);
      }
      XW_PH(2)
      __syncthreads();
      XW_PH(3)
      if (EDGE) {
#pragma unroll
        for (int pw = 0; pw < 4; ++pw) {
          const double* set = lds + pw * S::wset;
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            if ((ks & 1) == 0) asm volatile("" ::: "memory");
            const double xc = set[offC + 4 * ks], w0 = set[offW + 4 * ks], w1 = set[offW + XW_TTILE + 4 * ks];
            accH[0] = XW_MFMA(xc, w0, accH[0]);
            accH[1] = XW_MFMA(xc, w1, accH[1]);
          }
          asm volatile("" ::: "memory");
          const double x = set[offX];
#pragma unroll
          for (int e = 0; e < NE; ++e)
            if (e < NE - 1 || wave == 0) accE[e] = XW_MFMA4(x, set[offV + e * 4 * XW_TSTRIDE], accE[e]);
        }
      } else {
#pragma unroll
        for (int pw = 0; pw < 4; ++pw) {
          const double* setD = lds + S::oD + pw * S::wset;
          const double* setR = lds + S::oR + pw * S::wset;
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            if ((ks & 1) == 0) asm volatile("" ::: "memory");
            // TW x TW tiles of dVh per wave: 2 TW operand reads feed TW^2 MFMAs (2 x 2: four reads, four MFMAs; 4 x 4: eight, sixteen)
            double aa[TW], bb[TW];
#pragma unroll
            for (int i_ = 0; i_ < TW; ++i_) {
              aa[i_] = rec_readT<W>(setD, TW * wi + i_, ks);
              bb[i_] = rec_readT<W>(setR, TW * wj + i_, ks);
            }
#pragma unroll
            for (int i_ = 0; i_ < TW; ++i_)
#pragma unroll
              for (int j_ = 0; j_ < TW; ++j_) accH[i_ * TW + j_] = XW_MFMA(aa[i_], bb[j_], accH[i_ * TW + j_]);
          }
        }
      }
      XW_PH(4)
      __syncthreads();
      XW_PH(5)
#pragma unroll
      for (int mt = 0; mt < D::MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) dl[mt][r] = open[4 * mt + r] ? nd[mt][r] : 0.0;
    }
    // ---- dl = cotangent of a_0.  Input layer: dVin = dl (x) [t; x; 1], 48 input rows at a time
    // contraction of a cotangent tile set Dm (chain layout) with the input rows of this wave's 16 points over the block's 64
    // points; rows: 0 = all of [t; x; 1] (one unit), 1 = [0; x; 1] (a path group's sum over the time indices), 2 = [1; 0; 0]
    // (the time column from the t-weighted sum).  acc0: what tile 0 of group 0 starts from; the result goes to the slab.
    constexpr int RH = (D::MT + 3) / 4;       // row tiles of dVin a wave owns: wave, wave + 4, ... (below MT)
    struct InAcc { d4 v[RH]; };
    auto contract_input = [&](const d4 (&Dm)[D::MT], const int rows, const InAcc& acc0, InAcc* keep0) {
#pragma unroll
      for (int mt = 0; mt < D::MT; ++mt) {
        if (mt < D::MT - 1 || !S::VT) xw_writeT(myD + S::toff(mt), Dm[mt]);
        else xw_writeT_n<1>(myD + S::toff(mt), Dm[mt]);
      }
      const double* xl = xT;                    // laundered: the per-lane row addresses of x and of the slab are formed here,
      double* sl = slab;                        // not hoisted out of the tile loop into (spilled) registers
      int gl = g, nl = n;                       // (and copies of the lane coordinates the compiler cannot see through)
      asm volatile("" : "+s"(xl), "+s"(sl), "+v"(gl), "+v"(nl));
#pragma unroll
      for (int grp = 0; grp < NG; ++grp) {
        if (rows == 2 && grp > 0) break;
        const int nct = rows == 2 ? 1 : (d + 2 - 48 * grp >= 33 ? 3 : d + 2 - 48 * grp >= 17 ? 2 : 1);     // live 16-row tiles of this group
#pragma unroll
        for (int rr = 0; rr < 12; ++rr) {
          if (rr >= 4 * nct) continue;
          const int cl = gl + 4 * rr;           // local row 0..47 of this group
          const int c = 48 * grp + cl;          // input row: 0 = t, 1..d = x, d+1 = ones
          double val = 0.0;
          if (pt.valid) {
            if (rows == 2) val = c == 0 ? 1.0 : 0.0;
            else if (c == 0) val = rows == 0 ? pt.t : 0.0;
            else if (c <= d) val = xw_ld_g(xl + (long)(c - 1) * N + pt.n);
            else if (c == d + 1) val = 1.0;
          }
          myR[S::toff(cl >> 4) + (cl & 15) * XW_TSTRIDE + nl] = val;
        }
        __syncthreads();
        d4 accIn[RH][3];
#pragma unroll
        for (int rh = 0; rh < RH; ++rh) {
#pragma unroll
          for (int ct = 0; ct < 3; ++ct) accIn[rh][ct] = xw_zero4();
          if (grp == 0) accIn[rh][0] = acc0.v[rh];
        }
#pragma unroll
        for (int pw = 0; pw < 4; ++pw) {
          const double* setD = lds + S::oD + pw * S::wset;
          const double* setR = lds + S::oR + pw * S::wset;
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            double av[RH], bv[3];
#pragma unroll
            for (int rh = 0; rh < RH; ++rh) av[rh] = rec_readT<W>(setD, wave + 4 * rh < D::MT ? wave + 4 * rh : wave, ks);   // (a tile the wave does not own: a copy, never stored)
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) bv[ct] = ct < nct ? rec_readT<W>(setR, ct, ks) : 0.0;
#pragma unroll
            for (int rh = 0; rh < RH; ++rh)
#pragma unroll
              for (int ct = 0; ct < 3; ++ct)
                if (ct < nct) accIn[rh][ct] = XW_MFMA(av[rh], bv[ct], accIn[rh][ct]);
          }
        }
        __syncthreads();
        if (keep0 != nullptr) {                 // (the time-column pass: its tiles are what the next pass starts from)
#pragma unroll
          for (int rh = 0; rh < RH; ++rh) keep0->v[rh] = accIn[rh][0];
          break;
        }
        // dVin is touched once per contraction: it is accumulated in the block's slab (L2), not in 24 registers per group that
        // would be live through every layer above.  Wave `wave` owns rows [16 wave, 16 wave + 16) (+ 64 in the wide container).
#pragma unroll
        for (int rh = 0; rh < RH; ++rh)
#pragma unroll
          for (int ct = 0; ct < 3; ++ct) {
            if (ct >= nct) continue;
            const int c = 48 * grp + 16 * ct + nl;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = 16 * (wave + 4 * rh) + gl + 4 * r;
              if (wave + 4 * rh < D::MT && row < W && c <= d + 1) {
                double* dst = c <= d ? sl + o.Vin + row * o.ldin + c : sl + o.Vinb + row;
                xw_st_g(slab_first ? accIn[rh][ct][r] : xw_ld_g(dst) + accIn[rh][ct][r], dst);
              }
            }
          }
      }
    };
    InAcc zacc;
#pragma unroll
    for (int rh = 0; rh < RH; ++rh) zacc.v[rh] = xw_zero4();
    if (!TSUM) {
      contract_input(dl, 0, zacc, nullptr);
      slab_first = false;
    } else {
      const double tl = pt.t;                   // (one time index per unit)
#pragma unroll
      for (int mt = 0; mt < D::MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (r < D::LR(mt)) {
            sumS[mt][r] += dl[mt][r];
            sumT[mt][r] = fma(tl, dl[mt][r], sumT[mt][r]);
          }
      if (it + 1 == u1 || (it + 1) / L != ung) {           // (uniform) the path group ends here: contract its sums
        InAcc tcol;
        contract_input(sumT, 2, zacc, &tcol);
        contract_input(sumS, 1, tcol, nullptr);
        slab_first = false;
#pragma unroll
        for (int mt = 0; mt < D::MT; ++mt) sumS[mt] = sumT[mt] = xw_zero4();
      }
    }
    XW_PH(7)
  }
#ifdef XW_REC_PROBE
  const unsigned long long ck_loop = __builtin_amdgcn_s_memtime();
#endif

  if (EDGE) {
    // core tiles: (row band wave, column tile t) / wave 3 transposed: (row band t, column tile 2)
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wave < 3 ? 16 * wave + g + 4 * r : 16 * t2 + n, c = wave < 3 ? 16 * t2 + n : 32 + g + 4 * r;
        slab[o.Vh + row * W + c] = accH[t2][r];
      }
    // tile (2, 2): the four waves' partial sums, through the idle tile area
    __syncthreads();
    double* s22 = lds + S::oD;
#pragma unroll
    for (int r = 0; r < 4; ++r) s22[wave * 256 + (g + 4 * r) * 16 + n] = accH[2][r];
    __syncthreads();
    {
      const int i = threadIdx.x >> 4, j = threadIdx.x & 15;
      slab[o.Vh + (32 + i) * W + 32 + j] = (s22[threadIdx.x] + s22[256 + threadIdx.x]) + (s22[512 + threadIdx.x] + s22[768 + threadIdx.x]);
    }
    // edges: fold the four point groups (lane bits 2, 3); lane j + 16 i of group 0 holds entry (i, j) of the block
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      double s_ = accE[e];
      s_ += __shfl_xor(s_, 4);
      s_ += __shfl_xor(s_, 8);
      if ((e < NE - 1 || wave == 0) && (lane & 12) == 0) {
        const int i = lane >> 4, j = lane & 3, blk = vstart + e;
        // waves 0, 1: (cotangent row 48 + i, input row 4 blk + j); waves 2, 3 transposed: (input row 48 + i, cotangent row 4 blk + j)
        const int row = wave < 2 ? 48 + i : 4 * blk + j, c = wave < 2 ? 4 * blk + j : 48 + i;
        if (row < W) {
          if (c < W) slab[o.Vh + row * W + c] = s_;
          else if (c == W) slab[o.Vhb + row] = s_;
        }
      }
    }
  } else {
    // wave (wi, wj) owns row tiles TW wi .. x column tiles TW wj .. of dVh
#pragma unroll
    for (int t4 = 0; t4 < TW * TW; ++t4) {
      const int c = 16 * (TW * wj + (t4 % TW)) + n;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * (TW * wi + (t4 / TW)) + g + 4 * r;
        if (row < W) {
          if (c < W) slab[o.Vh + row * W + c] = accH[t4][r];
          else if (c == W) slab[o.Vhb + row] = accH[t4][r];
        }
      }
    }
  }
  __syncthreads();
  const int tid = threadIdx.x;
  if (!D::BIASROW) {   // dVh.b: the waves' sums over their points, [4 waves][16 MT rows] in the idle tile area
    double* sBsum = lds + S::oD;
#pragma unroll
    for (int mt = 0; mt < D::MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double s_ = xw_sum_over_n(accB[mt][r]);
        if (n == 0) sBsum[wave * 16 * D::MT + 16 * mt + 4 * r + g] = s_;
      }
    __syncthreads();
    if (tid < W) slab[o.Vhb + tid] = sBsum[tid] + sBsum[16 * D::MT + tid] + sBsum[32 * D::MT + tid] + sBsum[48 * D::MT + tid];
  }
  if (tid <= W) {
    double s_ = 0.0;
    if (tid < W) {
      const int mt = tid >> 4, gg = tid & 3, r = (tid & 15) >> 2;
      for (int wv = 0; wv < 4; ++wv) s_ += sO[(mt * 4 + r) * 16 + wv * 4 + gg];
      slab[o.Vo + tid] = s_;
    } else {
      for (int wv = 0; wv < 4; ++wv) s_ += sO[S::NO * 16 + wv * 4];
      slab[o.Vob] = s_;
    }
  }
#ifdef XW_REC_PROBE
  if (lane == 0 && blockIdx.x * 4 + wave < 2048) {
    unsigned long long* o_ = xw_rec_clock + 12 * (blockIdx.x * 4 + wave);
    for (int i = 0; i < 8; ++i) o_[i] = phc[i];
    const unsigned long long ck_out = __builtin_amdgcn_s_memtime();
    o_[8] = ck_loop - ck_in - (phc[0] + phc[1] + phc[2] + phc[3] + phc[4] + phc[5] + phc[6] + phc[7]);   // prologue
    o_[9] = ck_out - ck_loop;                                                                          // epilogue
    o_[10] = ck_out - ck_in;
    o_[11] = __builtin_amdgcn_s_memrealtime() - rt_in;
  }
#endif
}

int bwd_blocks(long P) {
  long nsuper = (P + 63) / 64;
  return (int)(nsuper < 512 ? nsuper : 512);     // two resident blocks per CU (k_disc_rec); k_disc_bwd grid-strides
}

}  // namespace

// compiled widths: 50 (the reference's YAML; all kernels) and 64 (container of the widths above 50: forward with the
// fused input gradient + reverse from the record, any depth; no recomputing reverse kernels)
static bool disc_width_ok(int W) { return W == 50 || W == 64 || W == 96 || W == 128; }
// (other widths up to 128: the generic path of xw_generic.hip, always from a record -- row-major there, [rows][columns])
extern "C" int xw_disc_act_rows(int W, int q) { return ((disc_width_ok(W) && q >= 0) || xwg_disc_ok(1, W, q)) ? (q + 1) * W : XW_E_DIMS; }

// x-projection of the input layer for the N paths of a group: xproj[r][n] = Vin[r, 1..d] x_n + Vin.b[r] for r < W, zero for the
// padding rows up to 64.  One thread per (path, four rows); 41 MFLOP at d = 100, N = 8192: a few microseconds.
namespace {
__global__ void __launch_bounds__(256) k_disc_xproj(const double* __restrict__ xT, const double* __restrict__ ph, int N, int d, int W,
                                                     double* __restrict__ xproj) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  const int r0 = blockIdx.y * 4;
  if (n >= N) return;
  const VOff o = v_offsets(d, W);
  const int r1 = r0 + 1 < W ? r0 + 1 : W - 1, r2 = r0 + 2 < W ? r0 + 2 : W - 1, r3 = r0 + 3 < W ? r0 + 3 : W - 1, rr = r0 < W ? r0 : W - 1;
  const double* w0 = ph + o.Vin + (long)rr * o.ldin + 1;
  const double* w1 = ph + o.Vin + (long)r1 * o.ldin + 1;
  const double* w2 = ph + o.Vin + (long)r2 * o.ldin + 1;
  const double* w3 = ph + o.Vin + (long)r3 * o.ldin + 1;
  double a0 = ph[o.Vinb + rr], a1 = ph[o.Vinb + r1], a2 = ph[o.Vinb + r2], a3 = ph[o.Vinb + r3];
#pragma unroll 4
  for (int i = 0; i < d; ++i) {
    const double x = xT[(long)i * N + n];
    a0 = fma(w0[i], x, a0);
    a1 = fma(w1[i], x, a1);
    a2 = fma(w2[i], x, a2);
    a3 = fma(w3[i], x, a3);
  }
  xproj[(long)r0 * N + n] = r0 < W ? a0 : 0.0;
  xproj[(long)(r0 + 1) * N + n] = r0 + 1 < W ? a1 : 0.0;
  xproj[(long)(r0 + 2) * N + n] = r0 + 2 < W ? a2 : 0.0;
  xproj[(long)(r0 + 3) * N + n] = r0 + 3 < W ? a3 : 0.0;
}
}  // namespace

extern "C" int xw_disc_xproj(const double* xT, const double* phi, int N, int d, int W, double* xproj, void* stream) {
  if (!xT || !phi || !xproj || N <= 0 || d <= 0) return XW_E_ARG;
  if (!disc_width_ok(W)) return XW_E_DIMS;
  // (the table has 16 MT rows: 64 for the widths 50 and 64, 96 / 128 for the wide containers; four rows per thread)
  hipLaunchKernelGGL(k_disc_xproj, dim3((N + 255) / 256, W > 96 ? 32 : W > 64 ? 24 : 16), dim3(256), 0, (hipStream_t)stream, xT, phi, N, d, W, xproj);
  return xw_launch_status();
}

extern "C" int xw_disc_fwd(const double* xT, const double* t, const double* tpp, const double* phi, int N, int L, int d,
                           int W, int q, double* v, double* vt, double* gxv, double* gtv, int ngrad, int max_blocks,
                           double* act, void* stream) {
  return xw_disc_fwd_xproj(xT, t, tpp, phi, N, L, d, W, q, v, vt, gxv, gtv, ngrad, max_blocks, act, nullptr, stream);
}

extern "C" int xw_disc_fwd_xproj(const double* xT, const double* t, const double* tpp, const double* phi, int N, int L, int d,
                                 int W, int q, double* v, double* vt, double* gxv, double* gtv, int ngrad, int max_blocks,
                                 double* act, const double* xproj, void* stream) {
  if (!xT || !phi || !v || N <= 0 || L <= 0 || d <= 0 || q < 0) return XW_E_ARG;
  if (!tpp && !t) return XW_E_ARG;
  if (tpp && L != 1) return XW_E_ARG;
  if (gxv && (ngrad <= 0 || (long)ngrad > (long)N * L || q > XW_QMAX)) return XW_E_ARG;
  if (!disc_width_ok(W)) return xwg_disc_fwd(xT, t, tpp, phi, N, L, d, W, q, v, vt, gxv, gtv, ngrad, act, stream);
  if (xproj != nullptr && tpp != nullptr) return XW_E_ARG;             // (point mode: every point has its own x)
  const long ntiles = ((long)N * L + 15) / 16;
  long blocks = (ntiles + 3) / 4;
  long cap = max_blocks > 0 ? max_blocks : 512;   // default: 2 blocks per CU resident (launch bounds), grid-stride over tiles
  if (blocks > cap) blocks = cap;
  if ((long)N * L * 4 >= (1L << 31)) return XW_E_ARG;                  // 32-bit lane offsets into the activation record
  // more than one round of tiles per wave: tickets (see k_disc_fwd).  Every launch takes the next queue slot; a captured
  // launch keeps its slot for all replays, and launches that can overlap in time never share one.
  static const bool dyn_on = [] { const char* e = getenv("XW_DISC_DYNAMIC"); return !(e && e[0] == '0'); }();
  static unsigned int* qbase = [] { void* p = nullptr; (void)hipGetSymbolAddress(&p, HIP_SYMBOL(xw_disc_queue)); return (unsigned int*)p; }();
  static std::atomic<unsigned int> next_slot[2];        // [0]: eager launches, [1]: launches recorded into a graph
  const bool dyn = dyn_on && qbase != nullptr && ntiles > 4 * blocks;
  unsigned int* queue = nullptr;
  if (dyn) {
    // a captured launch keeps its slot for every replay: the two kinds draw from separate halves of the slot array, so
    // that an eager launch (module calls between sub-steps, other streams) can never meet a replay on the same words
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)stream, &cs) != hipSuccess) cs = hipStreamCaptureStatusNone;
    const int half = cs == hipStreamCaptureStatusActive ? 1 : 0;
    const unsigned int slot = half * (XW_DISC_SLOTS / 2) + next_slot[half].fetch_add(1) % (XW_DISC_SLOTS / 2);
    queue = qbase + (size_t)slot * ((XW_DISC_NQ + 1) * XW_DISC_QSTRIDE);
  }
  // (Two 16-point tiles per wave at one wave per SIMD -- every A-fragment read feeding four matrix instructions, 360-410 registers,
  //  no scratch, results bit-identical -- was built and measured in round 6 and LOSES: 214.9 against 201.4 us without the gradient
  //  tiles, 273.5 against 220.4 us with them (static split over pairs); profiles/r06_disc_fwd_two_tiles_per_wave.txt.  Two waves of
  //  one tile each are the better way to fill a SIMD at this width; the kernel was removed again.)
  static const bool vin_on = [] { const char* e = getenv("XW_DISC_VIN_LDS"); return !(e && e[0] == '0'); }();
  // (W = 128: the 131 KB of Vh fragments leave no LDS for input-layer fragments -- the x-projection table, or global loads)
  const int vks = xproj != nullptr ? -1 : (!vin_on || W > 64) ? 0 : ((d + 3) / 4 <= XW_VIN_KS ? XW_VIN_KS : XW_VIN_KS_WIDE);
#define XW_DISC_FWD2(W_, ACT_, DYN_, VIN_)                                                                                \
  hipLaunchKernelGGL((k_disc_fwd<W_, ACT_, DYN_, VIN_>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, xT, t,  \
                     tpp, phi, N, L, d, q, v, vt, gxv, gtv, ngrad, act, queue, xproj)
#define XW_DISC_FWD(W_, ACT_, DYN_) do { if (vks < 0) XW_DISC_FWD2(W_, ACT_, DYN_, -1); else if (vks == XW_VIN_KS) XW_DISC_FWD2(W_, ACT_, DYN_, XW_VIN_KS); else if (vks) XW_DISC_FWD2(W_, ACT_, DYN_, XW_VIN_KS_WIDE); else XW_DISC_FWD2(W_, ACT_, DYN_, 0); } while (0)
#define XW_DISC_FWD_W(W_)                                                                                                 \
  if (act != nullptr) {                                                                                                   \
    if (dyn) XW_DISC_FWD(W_, true, true); else XW_DISC_FWD(W_, true, false);                                              \
  } else {                                                                                                                \
    if (dyn) XW_DISC_FWD(W_, false, true); else XW_DISC_FWD(W_, false, false);                                            \
  }
  if (W == 50) { XW_DISC_FWD_W(50) } else if (W == 64) { XW_DISC_FWD_W(64) }
  else {
    // 96, 128: one block per CU (the fragments of Vh are 74 / 131 KB of LDS, a wave holds up to 512 registers)
    if (blocks > 256) blocks = 256;
#define XW_DISC_FWDW(W_, ACT_, DYN_) do { if (vks < 0) XW_DISC_FWD2(W_, ACT_, DYN_, -1); else XW_DISC_FWD2(W_, ACT_, DYN_, 0); } while (0)
#define XW_DISC_FWD128(ACT_, DYN_) do { if (W == 96) XW_DISC_FWDW(96, ACT_, DYN_); else XW_DISC_FWDW(128, ACT_, DYN_); } while (0)
    const bool dyn128 = dyn && ntiles > 4 * blocks;
    if (!dyn128) queue = nullptr;
    if (act != nullptr) { if (dyn128) XW_DISC_FWD128(true, true); else XW_DISC_FWD128(true, false); }
    else { if (dyn128) XW_DISC_FWD128(false, true); else XW_DISC_FWD128(false, false); }
#undef XW_DISC_FWD128
#undef XW_DISC_FWDW
  }
#undef XW_DISC_FWD_W
#undef XW_DISC_FWD
#undef XW_DISC_FWD2
  return xw_launch_status();
}

#ifdef XW_CLOCK_PROBE
extern "C" int xw_debug_clock(unsigned long long* host, int nwaves) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(xw_clock_buf), sizeof(unsigned long long) * 2 * nwaves);
}
#endif

#ifdef XW_REC_PROBE
extern "C" int xw_debug_rec_clock(unsigned long long* host, int nwaves) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(xw_rec_clock), sizeof(unsigned long long) * 12 * nwaves);
}
#endif

extern "C" int xw_disc_bwd_slabs(int N, int L) { return bwd_blocks((long)N * L); }

#define XW_DISC_BWD(PARAMS, INGRAD)                                                                                      \
  if (d + 2 <= 64)                                                                                                       \
    hipLaunchKernelGGL((k_disc_bwd<50, 9, 1, PARAMS, INGRAD>), dim3(blocks), dim3(256), 0, s, xT, t, tpp, phi, vbar,     \
                       N, L, d, gslab, gxv, gtv);                                                                        \
  else                                                                                                                   \
    hipLaunchKernelGGL((k_disc_bwd<50, 9, 2, PARAMS, INGRAD>), dim3(blocks), dim3(256), 0, s, xT, t, tpp, phi, vbar,     \
                       N, L, d, gslab, gxv, gtv);

extern "C" int xw_disc_bwd(const double* xT, const double* t, const double* tpp, const double* phi, const double* vbar,
                           int N, int L, int d, int W, int q, const double* act, double* gslab, void* stream) {
  if (!xT || !phi || !gslab || N <= 0 || L <= 0 || d <= 0) return XW_E_ARG;
  if (!tpp && !t) return XW_E_ARG;
  if (tpp && L != 1) return XW_E_ARG;
  if (!disc_width_ok(W)) return xwg_disc_bwd(xT, t, tpp, phi, vbar, N, L, d, W, q, act, gslab, bwd_blocks((long)N * L), stream);
  // from the record: any depth (the layer loop is rolled -- unrolled for q = 9 it spilled and was 2 % slower); without a
  // record only the recomputing kernel of depth 9 (the reference's YAML) exists
  if (q < 0 || d + 2 > 128 || ((q != 9 || W != 50) && act == nullptr)) return XW_E_DIMS;
  if ((long)N * L * 4 >= (1L << 31)) return XW_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int blocks = bwd_blocks((long)N * L);
  double* gxv = nullptr;
  double* gtv = nullptr;
  if (act != nullptr) {
    const int ng = d + 2 <= 48 ? 1 : d + 2 <= 96 ? 2 : 3;
    // (vertical paths in whole 64-path groups: the input layer's gradient once per path group, see k_disc_rec)
    static const bool tsum_on = [] { const char* e = getenv("XW_DISC_REC_TSUM"); return !(e && e[0] == '0'); }();
    const bool tsum = tsum_on && tpp == nullptr && (N % 64) == 0;
#define XW_DISC_REC(W_, Q, NG)                                                                                          \
    if (tsum) hipLaunchKernelGGL((k_disc_rec<W_, Q, NG, true>), dim3(blocks), dim3(256), 0, s, xT, t, tpp, phi, vbar, N, L, d, gslab, act, q); \
    else hipLaunchKernelGGL((k_disc_rec<W_, Q, NG, false>), dim3(blocks), dim3(256), 0, s, xT, t, tpp, phi, vbar, N, L, d, gslab, act, q);
    if (W == 128) {
      if (ng == 1) { XW_DISC_REC(128, 0, 1) } else if (ng == 2) { XW_DISC_REC(128, 0, 2) } else { XW_DISC_REC(128, 0, 3) }
    } else if (W == 96) {
      if (ng == 1) { XW_DISC_REC(96, 0, 1) } else if (ng == 2) { XW_DISC_REC(96, 0, 2) } else { XW_DISC_REC(96, 0, 3) }
    } else if (W == 64) {
      if (ng == 1) { XW_DISC_REC(64, 0, 1) } else if (ng == 2) { XW_DISC_REC(64, 0, 2) } else { XW_DISC_REC(64, 0, 3) }
    } else {
      if (ng == 1) { XW_DISC_REC(50, 0, 1) } else if (ng == 2) { XW_DISC_REC(50, 0, 2) } else { XW_DISC_REC(50, 0, 3) }
    }
#undef XW_DISC_REC
  } else {
    XW_DISC_BWD(true, false)
  }
  return xw_launch_status();
}

extern "C" int xw_disc_gradx(const double* xT, const double* t, const double* tpp, const double* phi, const double* vbar,
                             int N, int d, int W, int q, double* gxv, double* gtv, void* stream) {
  if (!xT || !phi || !gxv || N <= 0 || d <= 0) return XW_E_ARG;
  if (!tpp && !t) return XW_E_ARG;
  if (W != 50 || q != 9 || d + 2 > 128) return XW_E_DIMS;
  hipStream_t s = (hipStream_t)stream;
  const int L = 1;
  const int blocks = bwd_blocks((long)N);
  double* gslab = nullptr;
  XW_DISC_BWD(false, true)
  return xw_launch_status();
}
