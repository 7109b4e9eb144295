// xw_ode_n4.h -- NARROW-TILE stepper kernels: one wave = 4 Monte-Carlo paths x 16 rows (included by xw_ode.hip, inside its
// anonymous namespace; same jobs, same activation store, same slabs, same results as the 16-path kernels).
//
// Why.  A 16-path tile is ONE instruction stream; N = 4096 paths are 256 of them for the chip's 1024 SIMDs, and a stream is
// a dependent chain (62 field evaluations x 9 layers per sweep): alone on the chip the 16-path sweeps run at 0.1 - 0.2 of
// the FP64 peak with three SIMDs in four idle (profiles/r03_sq_counters_ode.json: SQ_WAVES 256, matrix pipe 10 % busy).
// Here a block is still one 16-path tile (same grid, same slab, same activation-store tile) but its FOUR waves own four
// paths each, so a launch has four times the streams and each stream is shorter: a K x K layer is 4 dependent
// v_mfma_f64_4x4x4_4b instructions on ONE accumulator register instead of 9 on three.
//
// Layout.  v_mfma_f64_4x4x4_4b computes four independent 4x4x4 products ("blocks", lane bits 2..3):
//     A[blk][i][k] : lane = i + 4 blk + 16 k      B[blk][k][j] : lane = j + 4 blk + 16 k      D[blk][i][j] : lane = j + 4 blk + 16 i
// The 16-path kernels use the four blocks as four groups of 4 paths (same weight block in all of them).  Here the four
// blocks are four ROW blocks and the 4 columns j are the wave's 4 paths:
//     "C layout":  lane (lo = lane & 3, b = (lane >> 2) & 3, hi = lane >> 4)  holds  X[row 4 b + hi][path lo]
// -- a 16-row activation is ONE f64 register, what the instruction writes as D and (hi as the k index) reads as B.
// out = W in  needs block b of the output to meet all four blocks of the input:  for c = 0..3
//     D[b] += W[b][s_c(b)] in[s_c(b)],     s_c(b) = the block that lane block b sees after rotating the register by c blocks
// with the rotated operand made by two v_mov_b32 DPP row_ror:4c (no LDS, no ds_bpermute), and the weight operand of
// rotation c holding, in block b, the 4x4 weight block (b, s_c(b)).  s_c is MEASURED in the kernel (the block index itself is
// sent through the same DPP move), so nothing here depends on which way the hardware calls "right".
// Contractions over the PATHS (weight gradients) need operands with the path index in the k position:
//     "T layout":  lane (lo, b, hi)  holds  X[row 4 b + lo][path hi]
// A register goes from C to T with ONE matrix instruction (itself as A against a 4x4 identity as B: the instruction's
// own operand maps do the lane transposition, exactly -- x * 1 + 0 * ...); layer inputs are loaded from the activation
// store directly in T layout.  dW[4 b + hi][4 s_c(b) + lo] += MFMA(A = Q_T, B = rot_c(R_T)).
//
// Cost.  Per field evaluation of 4 paths: 41 chain instructions (+ 10 transposes + 38 outer products with weight
// gradients) of 18 clocks, i.e. ~1.8x the matrix-pipe time per PATH of the 16-path form (a quarter of every block product
// of a K = 10 layer is padding) -- so this layout is for launches that leave SIMDs idle (sweep B of a generator sub-step,
// small shards), not for phases bound by the sum of SIMD time.
//
// H > 16: the hidden state is two registers.  For 16 < H <= 20 the second one holds rows 16..19 REPLICATED in all four
// blocks: as an input it then needs one instruction instead of four, as an output the four rotations leave the complete sum
// in every block for free.  Wider (H <= 32): rows 16..31 as a second natural register.

namespace n4 {

template <int C> __device__ __forceinline__ int rot_i(int x) {
  if constexpr (C == 0) return x;
  else return __builtin_amdgcn_update_dpp(0, x, 0x120 + 4 * C, 0xf, 0xf, true);      // row_ror:4C
}
template <int C> __device__ __forceinline__ double rot(double x) {
  if constexpr (C == 0) return x;
  else return __hiloint2double(rot_i<C>(__double2hiint(x)), rot_i<C>(__double2loint(x)));
}
struct R4 { double v[4]; };
__device__ __forceinline__ R4 rots(double x) { return R4{{x, rot<1>(x), rot<2>(x), rot<3>(x)}}; }

struct Geo {
  int lo, b, hi;       // lane & 3, (lane >> 2) & 3, lane >> 4
  int sb[4];           // the block a lane of block b reads under rot<c>
  int q;               // wave of the block = which four of the tile's 16 paths
};
__device__ __forceinline__ Geo geo() {
  Geo g;
  const int l = xw_lane();
  g.lo = l & 3; g.b = (l >> 2) & 3; g.hi = l >> 4;
  g.sb[0] = g.b; g.sb[1] = rot_i<1>(g.b); g.sb[2] = rot_i<2>(g.b); g.sb[3] = rot_i<3>(g.b);
  g.q = (int)(threadIdx.x >> 6);
  return g;
}

template <int H, int K> struct Dn {
  static_assert(K <= 16 && H <= 32 && H % 4 == 0, "one register of pre-activations, at most two of hidden state");
  static constexpr int NY = H > 16 ? 2 : 1;
  static constexpr bool REP = H > 16 && H <= 20;       // second register: rows 16..19 in every block
  static constexpr int NC1 = NY == 2 ? (REP ? 1 : 4) : 0;   // instructions the second register needs as an INPUT
  static constexpr int KB = (K + 3) / 4, HB = (H + 3) / 4;
};

// A operand of  out[r0 ..] += Wm[.., c0 ..] in  for rotation c: lane (lo, b, hi) = Wm[r0 + 4 b + lo][c0 + 4 s_c(b) + hi], 0 outside
// Wm's R x C.  TR: Wm = src^T (src is C x R, leading dimension ld).  ROWS_REP / COLS_REP: the output / input register holds
// its four rows in every block (row r0 + lo resp. column c0 + hi, whatever the block).
template <bool TR, bool ROWS_REP = false, bool COLS_REP = false>
__device__ __forceinline__ double frag(const double* __restrict__ src, int ld, int R, int C, int r0, int c0, const Geo& g, int c) {
  const int r = r0 + (ROWS_REP ? 0 : 4 * g.b) + g.lo;
  const int cc = c0 + (COLS_REP ? 0 : 4 * g.sb[c]) + g.hi;
  if (!(r < R && cc < C)) return 0.0;
  return TR ? xw_ld_g(src + ((long)cc * ld + r)) : xw_ld_g(src + ((long)r * ld + cc));
}

// transposed weight operands of the adjoint chain
template <int H, int K> struct WT4 {
  double WoT0[4], WoT1[4], WhT[4], WyT0[4], WyT1[4];
};
template <int H, int K>
__device__ __forceinline__ void load_WT4(const double* __restrict__ th, const UOff& o, int d, const Geo& g, WT4<H, K>& w) {
  typedef Dn<H, K> D;
  const double* Wy = th + o.Win + d + 1;                  // Win[:, d+1:]  [K x H], leading dimension ldin
  constexpr int H0 = H < 16 ? H : 16;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    w.WoT0[c] = frag<true>(th + o.Wo, K, K, H0, 0, 0, g, c);                       // (Wo^T)[k][h] = Wo[h][k]
    w.WhT[c] = frag<true>(th + o.Wh, K, K, K, 0, 0, g, c);
    w.WyT0[c] = frag<true>(Wy, o.ldin, H0, K, 0, 0, g, c);                          // (Wy^T)[h][k] = Wy[k][h]
    w.WoT1[c] = 0.0;
    w.WyT1[c] = 0.0;
    if (D::NY == 2) {
      if (D::REP) {
        if (c == 0) w.WoT1[0] = frag<true, false, true>(th + o.Wo, K, K, H, 0, 16, g, 0);
        w.WyT1[c] = frag<true, true, false>(Wy, o.ldin, H, K, 16, 0, g, c);
      } else {
        w.WoT1[c] = frag<true>(th + o.Wo, K, K, H, 0, 16, g, c);
        w.WyT1[c] = frag<true>(Wy, o.ldin, H, K, 16, 0, g, c);
      }
    }
  }
}

// weight-gradient accumulators of a wave (T-layout products): element (row 4 b + hi, column 4 s_c(b) + lo) of register [c]
template <int H, int K> struct Acc4 {
  double Wh[4], Wo0[4], Wo1[4], Wy0[4], Wy1[4];
  double bo[Dn<H, K>::NY], bh, wt;          // C layout, summed over the paths at the end: Wo.b, Wh.b, the time column of Win
};

// lane offsets (doubles) into the activation store's tile of 16 paths (ActLayout; act_store() is what defines them)
template <int H, int K> struct Off4 {
  int cK, tK;                       // a K-row group: C layout / T layout
  int tY[Dn<H, K>::NY];             // an H-row group (stage inputs), T layout
  int word;                         // mask word of the lane's rows
};
template <int K> __device__ __forceinline__ int koff(int blk, int rowin, int n) {
  constexpr int KB = (K + 3) / 4;
  const int bk = blk < KB ? blk : KB - 1;                       // blocks past the last one: any finite copy
  const bool full = 4 * bk + 4 <= K;
  const int pk = K - 4 * bk;                                    // rows of a partial last block
  const int ri = full ? rowin : (rowin < pk ? rowin : pk - 1);
  return 64 * bk + (full ? 4 : pk) * n + ri;
}
template <int H, int K> __device__ __forceinline__ Off4<H, K> offsets(const Geo& g) {
  typedef Dn<H, K> D;
  Off4<H, K> f;
  f.cK = koff<K>(g.b, g.hi, 4 * g.q + g.lo);
  f.tK = koff<K>(g.b, g.lo, 4 * g.q + g.hi);
  const int b0 = g.b < D::HB ? g.b : D::HB - 1;
  f.tY[0] = 64 * b0 + 4 * (4 * g.q + g.hi) + g.lo;
  if (D::NY == 2) {
    const int b1 = D::REP ? 4 : (4 + g.b < D::HB ? 4 + g.b : D::HB - 1);
    f.tY[D::NY - 1] = 64 * b1 + 4 * (4 * g.q + g.hi) + g.lo;
  }
  f.word = 16 * g.hi + 4 * g.q + g.lo;
  return f;
}

// what a wave loads for one stage of one step
template <int H, int K, int M, bool PARAMS> struct Stage4 {
  unsigned word;                               // ReLU masks of the lane's rows as loaded: (layer j, block b) at bit base(j) - b
  double a;                                    // tanh(z_{m-1}), C layout
  double aT, rT[M > 1 ? M - 1 : 1];            // PARAMS: tanh and the layer inputs relu(z_j), T layout
  double yT[Dn<H, K>::NY];                     //         the stage input, T layout
  __device__ static constexpr int base(int j) { return Dn<H, K>::KB * (M - 1) - 1 - Dn<H, K>::KB * j; }
  // bits = word << (the lane's block): shifted where the stage is USED (a shift at the load would wait for the memory there)
  __device__ static __forceinline__ double gate(unsigned bits, int j, double x) {
    const int m = __builtin_amdgcn_sbfe((int)bits, base(j), 1);
    return __hiloint2double(__double2hiint(x) & m, __double2loint(x) & m);
  }
};

template <int H, int K, int M, int METHOD, bool PARAMS>
__device__ __forceinline__ void load_stage4(const double* __restrict__ Y, const double* __restrict__ act, int l, int i, int N,
                                            long ntile, int tile, const Geo& g, const Off4<H, K>& f, const long (&yT)[Dn<H, K>::NY],
                                            Stage4<H, K, M, PARAMS>& R) {
  typedef Dn<H, K> D;
  typedef ActLayout<H, K, M, RK<METHOD>::S> AL;
  static_assert(D::KB * (M - 1) + 3 <= 32, "mask word");
  const double* __restrict__ A = act + ((long)l * ntile + tile) * (AL::TOTAL * 16);
  typedef const unsigned __attribute__((address_space(1)))* gcu;
  R.word = __builtin_nontemporal_load((gcu)(reinterpret_cast<const unsigned*>(A + (AL::MASK + 2 * i) * 16) + f.word));
  const double* __restrict__ S = A + (long)i * AL::STAGE * 16;
  R.a = xw_ld_nt(S + (M - 1) * K * 16 + f.cK);
  if (PARAMS) {
    R.aT = xw_ld_nt(S + (M - 1) * K * 16 + f.tK);
#pragma unroll
    for (int j = 0; j < M - 1; ++j) R.rT[j] = xw_ld_nt(S + j * K * 16 + f.tK);
    if (i == 0) {
#pragma unroll
      for (int y = 0; y < D::NY; ++y) R.yT[y] = xw_ld_g(Y + (long)l * H * N + yT[y]);
    } else {
#pragma unroll
      for (int y = 0; y < D::NY; ++y) R.yT[y] = xw_ld_nt(A + (long)(AL::YI + (i - 1) * H) * 16 + f.tY[y]);
    }
  }
  __builtin_amdgcn_sched_barrier(0);                       // (requests stay where they are written: ahead of the stage in front)
}

// vector-Jacobian product of one field evaluation for 4 paths.  ob: cotangent of F's output (C layout); yb: cotangent of its
// y input; xpb += cotangent of z_0; PARAMS: the weight gradients of the evaluation go into G.
template <int H, int K, int M, bool PARAMS>
__device__ __forceinline__ void vjp4(const WT4<H, K>& w, double eye, int blk, double t, const Stage4<H, K, M, PARAMS>& s,
                                     const double (&ob)[Dn<H, K>::NY], double (&yb)[Dn<H, K>::NY], double& xpb, Acc4<H, K>& G) {
  typedef Dn<H, K> D;
  // nothing of this stage moves above this line: its operands were requested a stage ago, and a use that the scheduler
  // hoists to the request (it did: 1 - a^2 landed right behind the load of a) waits for the memory there
  __builtin_amdgcn_sched_barrier(0);
  const unsigned bits = s.word << blk;
  double ab = 0.0;
  {
    const R4 o0 = rots(ob[0]);
#pragma unroll
    for (int c = 0; c < 4; ++c) ab = XW_MFMA4(w.WoT0[c], o0.v[c], ab);
    if (D::NY == 2) {
      if (D::REP) ab = XW_MFMA4(w.WoT1[0], ob[D::NY - 1], ab);
      else {
        const R4 o1 = rots(ob[D::NY - 1]);
#pragma unroll
        for (int c = 0; c < 4; ++c) ab = XW_MFMA4(w.WoT1[c], o1.v[c], ab);
      }
    }
  }
  if (PARAMS) {
    const R4 aT = rots(s.aT);
    const double q0 = XW_MFMA4(ob[0], eye, 0.0);
#pragma unroll
    for (int c = 0; c < 4; ++c) G.Wo0[c] = XW_MFMA4(q0, aT.v[c], G.Wo0[c]);
    G.bo[0] += ob[0];
    if (D::NY == 2) {
      const double q1 = XW_MFMA4(ob[D::NY - 1], eye, 0.0);
      if (D::REP) G.Wo1[0] = XW_MFMA4(q1, aT.v[0], G.Wo1[0]);
      else {
#pragma unroll
        for (int c = 0; c < 4; ++c) G.Wo1[c] = XW_MFMA4(q1, aT.v[c], G.Wo1[c]);
      }
      G.bo[D::NY - 1] += ob[D::NY - 1];
    }
  }
  double zb = ab * fma(-s.a, s.a, 1.0);
#pragma unroll
  for (int j = M - 2; j >= 0; --j) {
    const R4 zr = rots(zb);
    if (PARAMS) {
      const double qz = XW_MFMA4(zb, eye, 0.0);
      const R4 rT = rots(s.rT[j]);
#pragma unroll
      for (int c = 0; c < 4; ++c) G.Wh[c] = XW_MFMA4(qz, rT.v[c], G.Wh[c]);
      G.bh += zb;
    }
    double tt = 0.0;
#pragma unroll
    for (int c = 0; c < 4; ++c) tt = XW_MFMA4(w.WhT[c], zr.v[c], tt);
    zb = Stage4<H, K, M, PARAMS>::gate(bits, j, tt);
  }
  xpb += zb;
  const R4 zr = rots(zb);
  if (PARAMS) {
    const double qz = XW_MFMA4(zb, eye, 0.0);
    const R4 y0 = rots(s.yT[0]);
#pragma unroll
    for (int c = 0; c < 4; ++c) G.Wy0[c] = XW_MFMA4(qz, y0.v[c], G.Wy0[c]);
    if (D::NY == 2) {
      if (D::REP) G.Wy1[0] = XW_MFMA4(qz, s.yT[D::NY - 1], G.Wy1[0]);
      else {
        const R4 y1 = rots(s.yT[D::NY - 1]);
#pragma unroll
        for (int c = 0; c < 4; ++c) G.Wy1[c] = XW_MFMA4(qz, y1.v[c], G.Wy1[c]);
      }
    }
    G.wt = fma(t, zb, G.wt);
  }
#pragma unroll
  for (int y = 0; y < D::NY; ++y) yb[y] = 0.0;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    yb[0] = XW_MFMA4(w.WyT0[c], zr.v[c], yb[0]);
    if (D::NY == 2) yb[D::NY - 1] = XW_MFMA4(w.WyT1[c], zr.v[c], yb[D::NY - 1]);
  }
}

// LDS plan of a block (doubles)
template <int H, int K> struct Plan4 {
  typedef Dn<H, K> D;
  static constexpr int NG = 12 + 2 * D::NC1;                       // T-product accumulators of a wave
  static constexpr int NTILE = 2 * D::NY + 2 + D::NY + 1;          // hand-over tiles: lam, accFL (NY each), xpb, bh, bo (NY), wt
  static constexpr int TILES = 0;                                  // XW_SWEEP_TILES tiles of the tail (wave 0)
  static constexpr int HAND = TILES + XW_SWEEP_TILES * XW_TTILE;   // NTILE tiles of 4 x 64: C-layout registers at their 16-path positions
  static constexpr int SCAL = HAND + NTILE * 256;                  // ub0[16], accFLb[16]
  static constexpr int GACC = SCAL + 32;                           // [4 waves][NG][64]
  static constexpr int TOTAL_P = GACC + 4 * NG * 64;
  static constexpr int TOTAL_X = SCAL + 32;
};

// a C-layout register of this wave -> its place in a chain-layout tile of the block's 16 paths (register r = block, lane
// 16 g + n with g = row in block, n = path): what the block's first wave reads back as a d4
__device__ __forceinline__ void hand_over(double* tile, const Geo& g, double x, bool rep) {
  if (!rep) tile[64 * g.b + 16 * g.hi + 4 * g.q + g.lo] = x;
  else tile[64 * g.b + 16 * g.hi + 4 * g.q + g.lo] = g.b == 0 ? x : 0.0;      // replicated rows: block 0 is the tile's register 0
}
__device__ __forceinline__ d4 take_over(const double* tile) {
  const int l = xw_lane();
  d4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = tile[64 * r + l];
  return v;
}
__device__ __forceinline__ void storeRowSumsStrided(double* dst, long stride, int rows, int r0, d4 q) {
  const int lane = xw_lane(), g = lane >> 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const double s = xw_sum_over_n(q[r]);
    const int row = r0 + g + 4 * r;
    if ((lane & 15) == 0 && row < rows) dst[(long)row * stride] = s;
  }
}

// the sweep: reverse of the discrete steps from the activation store (euler, midpoint), with or without weight gradients
template <int H, int K, int M, int METHOD, bool PARAMS>
__device__ __forceinline__ void sweep4(const BwdJobs& jobs, const double* __restrict__ tf, const double* __restrict__ th, int L, int d,
                                       double* lds, int vb) {
  typedef Dn<H, K> D;
  typedef Dim<H, K> DW;
  typedef RK<METHOD> T;
  typedef Plan4<H, K> P;
  static_assert(T::S <= 2, "the activation store is used by euler and midpoint");
  static_assert(D::NY == DW::HT, "registers of the hidden state = row tiles of the 16-path layout");
  xw_setprio(jobs.prio);
  const Geo g = geo();
  const int job = find_job(jobs, vb);
  const double* __restrict__ xT = jobs.xT[job];
  const double* __restrict__ start = jobs.start[job];
  const double* __restrict__ Y = jobs.Y[job];
  const double* __restrict__ act = jobs.act[job];
  const int N = jobs.N[job];
  const int tile = vb - jobs.tile0[job];
  const int base = tile * 16;
  const long ntile = (N + 15) >> 4;
  const int colC = base + 4 * g.q + g.lo, colT = base + 4 * g.q + g.hi;       // this lane's path in the two layouts
  const bool valid = colC < N;
  const int nclC = valid ? colC : N - 1, nclT = colT < N ? colT : N - 1;
  const UOff o = u_offsets(d, H, K);
  WT4<H, K> w;
  load_WT4<H, K>(th, o, d, g, w);
  const Off4<H, K> f = offsets<H, K>(g);
  const double eye = g.lo == g.hi ? 1.0 : 0.0;
  // rows of the hidden state this lane holds (C layout) / loads (T layout); past H: a finite copy of the last row
  int rowC[D::NY], rowT[D::NY];
  long yC[D::NY], yT[D::NY];
  double flw[D::NY];
#pragma unroll
  for (int y = 0; y < D::NY; ++y) {
    const int rc = y == 0 ? 4 * g.b + g.hi : (D::REP ? 16 + g.hi : 16 + 4 * g.b + g.hi);
    const int rt = y == 0 ? 4 * g.b + g.lo : (D::REP ? 16 + g.lo : 16 + 4 * g.b + g.lo);
    rowC[y] = rc;
    rowT[y] = rt;
    yC[y] = (long)(rc < H ? rc : H - 1) * N + nclC;
    yT[y] = (long)(rt < H ? rt : H - 1) * N + nclT;
    flw[y] = rc < H ? xw_ld_g(th + o.FLw + rc) : 0.0;
  }
  (void)rowT;

  Acc4<H, K> G;
#pragma unroll
  for (int c = 0; c < 4; ++c) G.Wh[c] = G.Wo0[c] = G.Wo1[c] = G.Wy0[c] = G.Wy1[c] = 0.0;
#pragma unroll
  for (int y = 0; y < D::NY; ++y) G.bo[y] = 0.0;
  G.bh = G.wt = 0.0;
  double accFL[D::NY], lam[D::NY];
#pragma unroll
  for (int y = 0; y < D::NY; ++y) accFL[y] = lam[y] = 0.0;
  double accFLb = 0.0, xpb = 0.0, ub0 = 0.0;

  const Cot4 cot = make_cot(jobs, job, N, nclC, valid, Y);
  auto load_y = [&](int l, double (&yl)[D::NY]) {
#pragma unroll
    for (int y = 0; y < D::NY; ++y) yl[y] = (PARAMS && rowC[y] < H) ? xw_ld_g(Y + (long)l * H * N + yC[y]) : 0.0;
  };
  auto readout = [&](const double (&yl)[D::NY], double ub) {
    ub0 = ub;
#pragma unroll
    for (int y = 0; y < D::NY; ++y) {
      lam[y] = fma(ub, flw[y], lam[y]);
      if (PARAMS) accFL[y] = fma(ub, yl[y], accFL[y]);
    }
    if (PARAMS) accFLb += ub;
  };
  double kb[T::S][D::NY], psum[D::NY];
  auto begin_step = [&](int l) {
    const double dt = tf[l + 1] - tf[l];
#pragma unroll
    for (int i = 0; i < T::S; ++i)
#pragma unroll
      for (int y = 0; y < D::NY; ++y) {
        kb[i][y] = (dt * T::b(i)) * lam[y];
        if (i == 0) psum[y] = 0.0;
      }
  };
  auto reverse_stage = [&](int l, int i, const Stage4<H, K, M, PARAMS>& s) {
    const double t0 = tf[l], dt = tf[l + 1] - t0;
    double psi[D::NY];
    vjp4<H, K, M, PARAMS>(w, eye, g.b, t0 + T::c(i) * dt, s, kb[i], psi, xpb, G);
#pragma unroll
    for (int y = 0; y < D::NY; ++y) {
      psum[y] += psi[y];
#pragma unroll
      for (int j = 0; j < T::S; ++j)
        if (j < i && T::a(i, j) != 0.0) kb[j][y] = fma(dt * T::a(i, j), psi[y], kb[j][y]);
    }
  };
  auto end_step = [&](const double (&yl)[D::NY], double ub) {
#pragma unroll
    for (int y = 0; y < D::NY; ++y) lam[y] += psum[y];
    readout(yl, ub);
  };

  auto run = [&](auto weak_tag) {
    constexpr bool WEAK = decltype(weak_tag)::value;
    Stage4<H, K, M, PARAMS> sa, sb;
    // (request order = completion order: the cotangent of the NEXT step is requested before the stage record, so that the
    //  wait in front of its use at the loop head leaves the record's loads in flight -- from the prologue as from the back edge)
    const CotRaw raw_last = cot_issue<WEAK>(cot, L - 1);
    CotRaw raw = cot_issue<WEAK>(cot, L > 1 ? L - 2 : 0);
    if (L > 1) load_stage4<H, K, M, METHOD, PARAMS>(Y, act, L - 2, T::S - 1, N, ntile, tile, g, f, yT, sa);
    {
      double yl[D::NY];
      load_y(L - 1, yl);
      readout(yl, cot_value<WEAK>(cot, raw_last, L - 1, L));
    }
    for (int l = L - 2; l >= 0; --l) {
      const double ub = cot_value<WEAK>(cot, raw, l, L);          // (requested a step ago)
      double yl[D::NY];
      load_y(l, yl);
      if constexpr (T::S == 2) {
        // stage 1 lives in sa, stage 0 in sb: each is requested while the other one is reversed
        load_stage4<H, K, M, METHOD, PARAMS>(Y, act, l, 0, N, ntile, tile, g, f, yT, sb);
        raw = cot_issue<WEAK>(cot, l > 0 ? l - 1 : 0);
        begin_step(l);
        reverse_stage(l, 1, sa);
        load_stage4<H, K, M, METHOD, PARAMS>(Y, act, l > 0 ? l - 1 : 0, 1, N, ntile, tile, g, f, yT, sa);
        reverse_stage(l, 0, sb);
      } else {
        sb = sa;
        raw = cot_issue<WEAK>(cot, l > 0 ? l - 1 : 0);
        load_stage4<H, K, M, METHOD, PARAMS>(Y, act, l > 0 ? l - 1 : 0, 0, N, ntile, tile, g, f, yT, sa);
        begin_step(l);
        reverse_stage(l, 0, sb);
      }
      end_step(yl, ub);
    }
  };
  if (cot.weak) run(std::true_type{});
  else run(std::false_type{});

  // ---- hand the per-path registers to the block's first wave (16-path chain layout), the gradients to a 4-wave sum ------
  double* hand = lds + P::HAND;
#pragma unroll
  for (int y = 0; y < D::NY; ++y) {
    hand_over(hand + (0 * D::NY + y) * 256, g, lam[y], y == 1 && D::REP);
    if (PARAMS) {
      hand_over(hand + (1 * D::NY + y) * 256, g, accFL[y], y == 1 && D::REP);
      hand_over(hand + (2 * D::NY + 2 + y) * 256, g, G.bo[y], y == 1 && D::REP);
    }
  }
  hand_over(hand + (2 * D::NY) * 256, g, xpb, false);
  if (PARAMS) {
    hand_over(hand + (2 * D::NY + 1) * 256, g, G.bh, false);
    hand_over(hand + (3 * D::NY + 2) * 256, g, G.wt, false);
  }
  if (g.b == 0 && g.hi == 0) {
    lds[P::SCAL + 4 * g.q + g.lo] = ub0;
    lds[P::SCAL + 16 + 4 * g.q + g.lo] = accFLb;
  }
  if (PARAMS) {
    double* mine = lds + P::GACC + g.q * (P::NG * 64) + xw_lane();
    int r = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) mine[64 * (r++)] = G.Wh[c];
#pragma unroll
    for (int c = 0; c < 4; ++c) mine[64 * (r++)] = G.Wo0[c];
#pragma unroll
    for (int c = 0; c < 4; ++c) mine[64 * (r++)] = G.Wy0[c];
#pragma unroll
    for (int c = 0; c < D::NC1; ++c) mine[64 * (r++)] = G.Wo1[c];
#pragma unroll
    for (int c = 0; c < D::NC1; ++c) mine[64 * (r++)] = G.Wy1[c];
  }
  __syncthreads();
  double* slab = PARAMS ? jobs.gslab[job] + (long)tile * o.total : nullptr;
  if (PARAMS) {
    // register r of the block's sum is formed and stored by wave r mod 4: element (row 4 b + hi, column 4 s_c(b) + lo)
    const double* all = lds + P::GACC + xw_lane();
    const int row = 4 * g.b + g.hi;
#pragma unroll
    for (int r = 0; r < P::NG; ++r) {
      if ((r & 3) != g.q) continue;
      const double x = (all[64 * r] + all[64 * (P::NG + r)]) + (all[64 * (2 * P::NG + r)] + all[64 * (3 * P::NG + r)]);
      const int kind = r < 12 ? r >> 2 : 3 + (r - 12) / (D::NC1 > 0 ? D::NC1 : 1);      // 0 Wh, 1 Wo0, 2 Wy0, 3 Wo1, 4 Wy1
      const int c = r < 12 ? r & 3 : (r - 12) % (D::NC1 > 0 ? D::NC1 : 1);
      const int col = 4 * g.sb[c] + g.lo;
      if (kind == 0) {
        if (row < K && col < K) slab[o.Wh + row * K + col] = x;
      } else if (kind == 1) {
        if (row < H && col < K) slab[o.Wo + row * K + col] = x;
      } else if (kind == 2) {
        if (row < K && col < H) slab[o.Win + row * o.ldin + d + 1 + col] = x;
      } else if (kind == 3) {
        const int rw = D::REP ? 16 + g.hi : 16 + row;
        if (rw < H && col < K) slab[o.Wo + rw * K + col] = x;
      } else {
        const int cl = D::REP ? 16 + g.lo : 16 + col;
        if (row < K && cl < H) slab[o.Win + row * o.ldin + d + 1 + cl] = x;
      }
    }
  }
  if (g.q != 0) return;
  // ---- first wave: the tile's 16 paths in the chain layout, the common end of a sweep ------------------------------------
  {
    const int lane = xw_lane(), n = lane & 15;
    const bool valid16 = base + n < N;
    const int ncl16 = valid16 ? base + n : N - 1;
    d4 lam16[DW::HT], fl16[DW::HT], flw16[DW::HT];
#pragma unroll
    for (int ht = 0; ht < DW::HT; ++ht) {
      lam16[ht] = take_over(hand + (0 * D::NY + ht) * 256);
      fl16[ht] = PARAMS ? take_over(hand + (1 * D::NY + ht) * 256) : xw_zero4();
      flw16[ht] = xw_vecD(th + o.FLw, H, 16 * ht);
    }
    const d4 xpb16 = take_over(hand + (2 * D::NY) * 256);
    const double ub016 = lds[P::SCAL + n], flb16 = lds[P::SCAL + 16 + n];
    if (PARAMS) {
      storeRowSums(slab + o.Whb, K, 0, take_over(hand + (2 * D::NY + 1) * 256));
#pragma unroll
      for (int ht = 0; ht < DW::HT; ++ht) storeRowSums(slab + o.Wob, H, 16 * ht, take_over(hand + (2 * D::NY + 2 + ht) * 256));
      storeRowSumsStrided(slab + o.Win + d, o.ldin, K, 0, take_over(hand + (3 * D::NY + 2) * 256));
    }
    sweep_tail<H, K, PARAMS, false>(th, o, d, N, base, valid16, ncl16, xT, start[ncl16], jobs.x_ones != 0, lam16, xpb16, ub016, fl16,
                                    flb16, flw16, jobs.gx[job], jobs.gs[job], slab, lds + P::TILES, []() {});
  }
}

template <int H, int K, int M, int METHOD, bool PARAMS>
__global__ void __launch_bounds__(256, 2) k_ode_bwd_n4(const BwdJobs jobs, const double* __restrict__ tf,
                                                       const double* __restrict__ th, int L, int d) {
  __shared__ double lds[PARAMS ? Plan4<H, K>::TOTAL_P : Plan4<H, K>::TOTAL_X];
  sweep4<H, K, M, METHOD, PARAMS>(jobs, tf, th, L, d, lds, (int)blockIdx.x);
}

// ---- forward pass on narrow tiles -------------------------------------------------------------------------------------
// The same outputs as k_ode_fwd (u, the checkpoints Y, the activation store in ActLayout -- so that either kind of sweep
// can read what either kind of forward pass wrote), four waves of 4 paths per 16-path tile.  Per field evaluation 41 matrix
// instructions (5 + 4 (m - 1) + 8 at (H, K) = (20, 10)), tanh once per 16 rows x 4 paths, one ReLU / mask push per layer.
template <int H, int K> struct W4 {          // forward operands (rotation c): blocks (b, s_c(b)) of Wy, Wh, Wo
  double Wy0[4], Wy1[4], Wh[4], Wo0[4], Wo1[4];
  double wt, bh, bo[Dn<H, K>::NY];           // C layout: time column of Win, Wh.b, Wo.b
};
template <int H, int K>
__device__ __forceinline__ void load_W4(const double* __restrict__ th, const UOff& o, int d, const Geo& g, W4<H, K>& w) {
  typedef Dn<H, K> D;
  const double* Wy = th + o.Win + d + 1;
  constexpr int H0 = H < 16 ? H : 16;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    w.Wy0[c] = frag<false>(Wy, o.ldin, K, H0, 0, 0, g, c);
    w.Wh[c] = frag<false>(th + o.Wh, K, K, K, 0, 0, g, c);
    w.Wo0[c] = frag<false>(th + o.Wo, K, H0, K, 0, 0, g, c);
    w.Wy1[c] = 0.0;
    w.Wo1[c] = 0.0;
    if (D::NY == 2) {
      if (D::REP) {
        if (c == 0) w.Wy1[0] = frag<false, false, true>(Wy, o.ldin, K, H, 0, 16, g, 0);
        w.Wo1[c] = frag<false, true, false>(th + o.Wo, K, H, K, 16, 0, g, c);
      } else {
        w.Wy1[c] = frag<false>(Wy, o.ldin, K, H, 0, 16, g, c);
        w.Wo1[c] = frag<false>(th + o.Wo, K, H, K, 16, 0, g, c);
      }
    }
  }
  const int rk = 4 * g.b + g.hi;
  w.wt = rk < K ? xw_ld_g(th + o.Win + (long)rk * o.ldin + d) : 0.0;
  w.bh = rk < K ? xw_ld_g(th + o.Whb + rk) : 0.0;
#pragma unroll
  for (int y = 0; y < D::NY; ++y) {
    const int rh = y == 0 ? rk : (D::REP ? 16 + g.hi : 16 + rk);
    w.bo[y] = rh < H ? xw_ld_g(th + o.Wob + rh) : 0.0;
  }
}
// out[H] += Wm[H x H] in[H] in the C layout (the initial layers; operands straight from memory, used once)
template <int H, int K>
__device__ __forceinline__ void mat_hh(const double* __restrict__ Wm, const Geo& g, const double (&in)[Dn<H, K>::NY],
                                       double (&out)[Dn<H, K>::NY]) {
  typedef Dn<H, K> D;
  constexpr int H0 = H < 16 ? H : 16;
  const R4 r0 = rots(in[0]);
#pragma unroll
  for (int c = 0; c < 4; ++c) out[0] = XW_MFMA4(frag<false>(Wm, H, H0, H0, 0, 0, g, c), r0.v[c], out[0]);
  if (D::NY == 2) {
    if (D::REP) {
      out[0] = XW_MFMA4((frag<false, false, true>(Wm, H, H0, H, 0, 16, g, 0)), in[D::NY - 1], out[0]);
#pragma unroll
      for (int c = 0; c < 4; ++c) out[D::NY - 1] = XW_MFMA4((frag<false, true, false>(Wm, H, H, H0, 16, 0, g, c)), r0.v[c], out[D::NY - 1]);
      out[D::NY - 1] = XW_MFMA4((frag<false, true, true>(Wm, H, H, H, 16, 16, g, 0)), in[D::NY - 1], out[D::NY - 1]);
    } else {
      const R4 r1 = rots(in[D::NY - 1]);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        out[0] = XW_MFMA4(frag<false>(Wm, H, H0, H, 0, 16, g, c), r1.v[c], out[0]);
        out[D::NY - 1] = XW_MFMA4(frag<false>(Wm, H, H, H0, 16, 0, g, c), r0.v[c], out[D::NY - 1]);
        out[D::NY - 1] = XW_MFMA4(frag<false>(Wm, H, H, H, 16, 16, g, c), r1.v[c], out[D::NY - 1]);
      }
    }
  }
}
// sum over the 16 rows a lane group (lo fixed) holds: the four blocks of a 16-lane row, then the four rows of lanes
__device__ __forceinline__ double sum_rows(double x) {
  x += rot<2>(x);
  x += rot<1>(x);
  return xw_sum_over_g(x);
}

template <int H, int K, int M, int METHOD, int ACT>
__global__ void __launch_bounds__(256, 2) k_ode_fwd_n4(const FwdJobs jobs, const double* __restrict__ tf,
                                                       const double* __restrict__ th, int L, int d) {
  typedef Dn<H, K> D;
  typedef RK<METHOD> T;
  typedef ActLayout<H, K, M, T::S> AL;
  static_assert(D::KB * (M - 1) + 3 <= 32, "mask word");
  xw_setprio(jobs.prio);
  const Geo g = geo();
  const int job = find_job(jobs, (int)blockIdx.x);
  const double* __restrict__ xT = jobs.xT[job];
  double* __restrict__ u = jobs.u[job];
  double* __restrict__ Y = jobs.Y[job];
  double* __restrict__ act = jobs.act[job];
  const int N = jobs.N[job];
  const int tile = (int)blockIdx.x - jobs.tile0[job];
  const int col = tile * 16 + 4 * g.q + g.lo;
  const bool valid = col < N;
  const int ncl = valid ? col : N - 1;
  if (blockIdx.x == 0 && jobs.zero16 != nullptr && threadIdx.x < 16) jobs.zero16[threadIdx.x] = 0.0;
  const UOff o = u_offsets(d, H, K);
  W4<H, K> w;
  load_W4<H, K>(th, o, d, g, w);
  int rowC[D::NY];
  bool liveC[D::NY];                          // this lane holds a row of the hidden state that exists (and, REP: is block 0's copy)
  double flw[D::NY];
#pragma unroll
  for (int y = 0; y < D::NY; ++y) {
    rowC[y] = y == 0 ? 4 * g.b + g.hi : (D::REP ? 16 + g.hi : 16 + 4 * g.b + g.hi);
    liveC[y] = rowC[y] < H && !(y == 1 && D::REP && g.b != 0);
    flw[y] = liveC[y] ? xw_ld_g(th + o.FLw + rowC[y]) : 0.0;      // (a replicated row enters the read-out once)
  }
  const double flb = th[o.FLb];
  // ---- lift: start scalar -> y_0 (initial_layers of src/model.py:78,97) ---------------------------------------------------
  double y[D::NY];
  {
    const double sv = jobs.start[job][ncl];
    double a0[D::NY], a1[D::NY];
#pragma unroll
    for (int q = 0; q < D::NY; ++q) {
      const bool in = rowC[q] < H;
      a0[q] = in ? xw_relu1(fma(xw_ld_g(th + o.IL0w + rowC[q]), sv, xw_ld_g(th + o.IL0b + rowC[q]))) : 0.0;
      a1[q] = in ? xw_ld_g(th + o.IL2b + rowC[q]) : 0.0;
      y[q] = in ? xw_ld_g(th + o.IL4b + rowC[q]) : 0.0;
    }
    mat_hh<H, K>(th + o.IL2w, g, a0, a1);
#pragma unroll
    for (int q = 0; q < D::NY; ++q) a1[q] = xw_relu1(a1[q]);
    mat_hh<H, K>(th + o.IL4w, g, a1, y);
  }
  // ---- xp = Win.b + Win[:, :d] x  (time-invariant along a path) -------------------------------------------------------------
  double xp;
  {
    const int rk = 4 * g.b + g.hi;
    xp = rk < K ? xw_ld_g(th + o.Winb + rk) : 0.0;
    for (int r = 0; r < (d + 15) / 16; ++r) {
      const int dim = 16 * r + rk;
      const R4 xr = rots(dim < d ? xw_ld_g(xT + (long)dim * N + ncl) : 0.0);
#pragma unroll
      for (int c = 0; c < 4; ++c) xp = XW_MFMA4(frag<false>(th + o.Win, o.ldin, K, d, 0, 16 * r, g, c), xr.v[c], xp);
    }
  }
  // lane offsets (bytes) into the activation store's tile (C layout).  Every store of the time loop goes through a buffer
  // descriptor and lanes that own nothing carry an out-of-range offset (ActLane.off_part_st in xw_ode.hip: a branch on a lane
  // predicate around one store per layer cost more than the layer's four matrix instructions)
  const unsigned cK = 4 * g.b + g.hi < K ? 8u * (unsigned)koff<K>(g.b, g.hi, 4 * g.q + g.lo) : XW_ACT_OOB;
  unsigned cY[D::NY], oY[D::NY];
  const bool big = (long)H * N * 8 >= (1L << 31);         // (beyond a descriptor's 32-bit range: plain guarded stores)
#pragma unroll
  for (int q = 0; q < D::NY; ++q) {
    cY[q] = liveC[q] ? 8u * (unsigned)(64 * (q == 0 ? g.b : (D::REP ? 4 : 4 + g.b)) + 4 * (4 * g.q + g.lo) + g.hi) : XW_ACT_OOB;
    oY[q] = (valid && liveC[q]) ? 8u * (unsigned)(rowC[q] * N + col) : XW_ACT_OOB;
  }
  const unsigned oU = (g.b == 0 && g.hi == 0 && valid) ? 8u * (unsigned)col : XW_ACT_OOB;
  const unsigned oW = g.b == 0 ? 4u * (unsigned)(16 * g.hi + 4 * g.q + g.lo) : XW_ACT_OOB;
  const long ntile = (N + 15) >> 4;
#define st64(x, rs, voff, soff, aux)                                                                        \
  {                                                                                                         \
    const double x_ = (x);                                                                                  \
    const xw_u2v w2_ = {(unsigned)__double2loint(x_), (unsigned)__double2hiint(x_)};                        \
    __builtin_amdgcn_raw_buffer_store_b64(w2_, (rs), (int)(voff), (soff), (aux));                           \
  }

  // one field evaluation: F([x, t, y]) of src/model.py:153-156; S: the stage's part of the activation record (or nullptr)
  auto field = [&](double t, const double (&yi)[D::NY], double (&out)[D::NY], const __amdgpu_buffer_rsrc_t& rs, int i) {
    double z = fma(w.wt, t, xp);
    {
      const R4 r0 = rots(yi[0]);
#pragma unroll
      for (int c = 0; c < 4; ++c) z = XW_MFMA4(w.Wy0[c], r0.v[c], z);
      if (D::NY == 2) {
        if (D::REP) z = XW_MFMA4(w.Wy1[0], yi[D::NY - 1], z);
        else {
          const R4 r1 = rots(yi[D::NY - 1]);
#pragma unroll
          for (int c = 0; c < 4; ++c) z = XW_MFMA4(w.Wy1[c], r1.v[c], z);
        }
      }
    }
    unsigned bits = 0;
#pragma unroll
    for (int j = 0; j < M - 1; ++j) {
      // (the bit is z > 0, not the sign: a dead layer feeds exact +0 to the next one and relu'(+0) = 0, xw_ode.hip SaveX)
      bits = (bits << D::KB) | (z > 0.0 ? 1u : 0u);
      const double r = xw_relu1(z);
      if (ACT == 1) st64(r, rs, cK, (i * AL::STAGE + j * K) * 16 * 8, 2);      // (nt: streamed, read once, by a sweep)
      const R4 rr = rots(r);
      z = w.bh;
#pragma unroll
      for (int c = 0; c < 4; ++c) z = XW_MFMA4(w.Wh[c], rr.v[c], z);
    }
    const double a = xw_tanh(z);
    if (ACT) st64(a, rs, cK, (i * AL::STAGE + (M - 1) * K) * 16 * 8, 2);
    if (ACT) {
      // this lane pushed (layer j) at bit KB (M - 2 - j); the record wants (layer j, block b) at bit KB (M - 1) - 1 - KB j - b,
      // all blocks of a lane row in ONE word: shift by the block, OR over the four blocks, block 0 stores
      unsigned wd = bits << (D::KB - 1 - (g.b < D::KB ? g.b : D::KB - 1));
      wd = g.b < D::KB ? wd : 0u;
      wd |= (unsigned)rot_i<2>((int)wd);
      wd |= (unsigned)rot_i<1>((int)wd);
      __builtin_amdgcn_raw_buffer_store_b32(wd, rs, (int)oW, (AL::MASK + 2 * i) * 16 * 8, 2);
    }
    const R4 ar = rots(a);
#pragma unroll
    for (int q = 0; q < D::NY; ++q) out[q] = w.bo[q];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      out[0] = XW_MFMA4(w.Wo0[c], ar.v[c], out[0]);
      if (D::NY == 2) out[D::NY - 1] = XW_MFMA4(w.Wo1[c], ar.v[c], out[D::NY - 1]);
    }
  };

  for (int l = 0; l < L; ++l) {
    double part = 0.0;
    const bool ybuf = Y != nullptr && !big;
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(ybuf ? Y + (long)l * H * N : nullptr, 0, ybuf ? H * N * 8 : 0, 0x00020000);
#pragma unroll
    for (int q = 0; q < D::NY; ++q) {
      part = fma(flw[q], y[q], part);
      st64(y[q], yrs, oY[q], 0, 0);
    }
    if (big && Y != nullptr) {                            // (uniform, never taken below 13 million paths)
#pragma unroll
      for (int q = 0; q < D::NY; ++q)
        if (valid && liveC[q]) Y[((long)l * H + rowC[q]) * N + col] = y[q];
    }
    const double ul = sum_rows(part) + flb;               // final_linear, src/model.py:110
    st64(ul, __builtin_amdgcn_make_buffer_rsrc(u + (long)l * N, 0, N * 8, 0x00020000), oU, 0, 0);
    if (l == L - 1) break;
    const double t0 = tf[l], dt = tf[l + 1] - t0;
    double* __restrict__ A = ACT ? act + ((long)l * ntile + tile) * (AL::TOTAL * 16) : nullptr;
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(A, 0, ACT ? AL::TOTAL * 16 * 8 : 0, 0x00020000);
    double k[T::S][D::NY];
#pragma unroll
    for (int i = 0; i < T::S; ++i) {
      double yi[D::NY];
#pragma unroll
      for (int q = 0; q < D::NY; ++q) {
        yi[q] = y[q];
#pragma unroll
        for (int j = 0; j < i; ++j)
          if (T::a(i, j) != 0.0) yi[q] = fma(dt * T::a(i, j), k[j][q], yi[q]);
        if (ACT == 1 && i > 0) st64(yi[q], ars, cY[q], (AL::YI + (i - 1) * H) * 16 * 8, 2);
      }
      field(t0 + T::c(i) * dt, yi, k[i], ars, i);
    }
#pragma unroll
    for (int i = 0; i < T::S; ++i)
      if (T::b(i) != 0.0)
#pragma unroll
        for (int q = 0; q < D::NY; ++q) y[q] = fma(dt * T::b(i), k[i][q], y[q]);
  }
}

#undef st64
}  // namespace n4
