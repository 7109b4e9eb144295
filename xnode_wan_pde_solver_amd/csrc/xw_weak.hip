// xw_weak.hip -- weak-form functional, penalties, cotangents and the fused Adam update on gfx950.
//
// Replaces loss.I / init / bdry / int / u / v (src/loss.py:46-96 of the reference) and torch.optim.Adam.step
// (src/training.py:103-104,138,162).  These are HBM-streaming kernels over the time-major [L, N] point arrays:
// one lane per sample point (consecutive lanes = consecutive paths of one time index), partial sums are reduced
// wave -> block with shuffles and leave the block as one atomic per scalar.
//
// Reference semantics kept on purpose: s1 uses v, not phi; +f*phi; the u-factor of the
// d(phi)/dt term and the whole gradient-contraction term carry no gradient (Q2) -- they only enter the VALUE of I;
// nabla_x u only exists at the first time index (Q3), so the a_ij d_i phi d_j u contraction arrives pre-contracted
// per path in s3x[N].
#include "xw_common.h"
#include "xnwan.h"

namespace {

// sum over the 64 lanes, every lane gets the total: rows of 16 on DPP row operations, the four rows on lane swaps
// (xw_common.h) -- no ds_bpermute round trips (6 x 2 per value before; the reductions take five values per block)
__device__ __forceinline__ double wave_sum(double x) { return xw_sum_over_g(xw_sum_over_n(x)); }
// Deterministic grid-wide sum of NV per-thread partials: every block stores its partial sums to `work`, the block that
// arrives last (agent-scope ticket) adds them up in block order and accumulates into dst[0..NV).  Float atomics would
// be shorter but make two runs of the same step differ in the last bits; training must be bit-reproducible
// (checkpoint/resume, test_test_net_reuse_is_exact).  work: NV * gridDim.x doubles + 1 ticket word (kept at zero
// between launches: the last block resets it).  Hand-off follows cdna_hip_programming.md Guideline 16: stores ->
// s_waitcnt vmcnt(0) -> barrier -> lane-0 release fence -> ticket; consumer: ticket -> acquire fence -> barrier -> loads.
// NV > 3: values 3, 4 go to dst[7], dst[8] (scal[4..6] are the loss values), value 5 to dst[3] (the boundary sum of squares)
template <int NV>
__device__ __forceinline__ bool grid_sum(double (&val)[NV], double* __restrict__ work, double* __restrict__ dst) {
  __shared__ double red[NV][16];                        // (blocks of up to 16 waves)
  __shared__ int is_last;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const double s = wave_sum(val[i]);
    if (lane == 0) red[i][wave] = s;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    double s = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[threadIdx.x][w];
    work[(long)blockIdx.x * NV + threadIdx.x] = s;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  unsigned int* ticket = reinterpret_cast<unsigned int*>(work + (long)gridDim.x * NV);
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = (t == gridDim.x - 1);
    if (is_last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  if (!is_last) return false;
  // final sum by the whole last block in a fixed order: strided per-thread partials, then the same shuffle / LDS tree
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    double part = 0.0;
    for (unsigned int b = threadIdx.x; b < gridDim.x; b += blockDim.x) part += work[(long)b * NV + i];
    const double s = wave_sum(part);
    if (lane == 0) red[i][wave] = s;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    double tot = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += red[threadIdx.x][w];
    dst[threadIdx.x < 3 ? threadIdx.x : (threadIdx.x < 5 ? threadIdx.x + 4 : threadIdx.x - 2)] += tot;   // 0..2 | 7, 8 | 3
  }
  if (threadIdx.x == 0) *ticket = 0u;
  return true;                                          // this block completed the sum
}

__device__ __forceinline__ double interior_loss(const double* scal, double Vol, double Nglob, int L) {
  const double I = scal[0], S = scal[1];
  return log(I * I) - log(Vol * S / (Nglob * (double)L));  // src/loss.py:89-90
}

// Single-slice groups at T0 of the list domains: NeuralODE.forward returns [N,1] there instead of [N,1,1]
// (src/model.py:89-91) and the products of src/loss.py:65,70 broadcast to [N,N] tables over all PAIRS of paths, summed over
// both axes.  The sums factorise: sum_mn u_n dphi0_m = (sum u)(sum dphi0) -- the two factors are accumulated separately
// (scal[7], scal[8]) and folded into I here, once they are global (after the all-reduce on several GPUs).
__device__ __forceinline__ void fold_pairs(double* scal, double cN) {
  scal[0] -= cN * scal[7] * scal[8];
  scal[7] = 0.0;
  scal[8] = 0.0;
}
__device__ __forceinline__ void loss_values(double* scal, int L, int Lb, double Vol, double Nglob, double Nbglob,
                                            double alpha, double init_off, double bdry_off) {
  const double in_ = interior_loss(scal, Vol, Nglob, L);
  scal[6] = in_;
  // init_off / bdry_off: the sample-only part of a pairwise mean, mean_nm (u_n - h_m)^2 = mean_n (u_n - mean h)^2 + var h
  scal[4] = in_ + alpha * ((scal[2] / Nglob + init_off) + (scal[3] / (Nbglob * (double)Lb) + bdry_off));  // src/loss.py:93
  scal[5] = -in_;                                                                                          // src/loss.py:96
}

__global__ void __launch_bounds__(1024) k_weak_partials(const double* __restrict__ u, const double* __restrict__ v,
                                                       const double* __restrict__ vt, const double* __restrict__ w,
                                                       int w_per_point, const double* __restrict__ wt,
                                                       const double* __restrict__ s3x, const double* __restrict__ gx,
                                                       const double* __restrict__ gs, const double* __restrict__ ghT,
                                                       const double* __restrict__ gxv, const double* __restrict__ w0,
                                                       const double* __restrict__ gwx0T, int d,
                                                       const double* __restrict__ c, double ckappa,
                                                       const double* __restrict__ f, const double* __restrict__ h,
                                                       const double* __restrict__ href, int pairwise, double s3_scale,
                                                       int N, int L, double Vol, double Nglob, double* __restrict__ work,
                                                       double* __restrict__ scal, int finalize, int Lb, double Nbglob,
                                                       double alpha, double init_off, double bdry_off,
                                                       long long* __restrict__ step, const double* __restrict__ ub,
                                                       const double* __restrict__ gb, long Pb) {
  // one lane per sample point (time-major: consecutive lanes = consecutive paths of one time index, coalesced)
  // pairwise (single-slice T0 group, L == 1): the s2 term is (sum u)(sum dphi0) -- factors in acc[3], acc[4] -- and the
  // caller passes s3_scale = N with f := mean f, href := mean h (the [N,N] sums of src/loss.py:70,79 factorised)
  double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};  // I, sum v^2, SSE_init, sum u, sum dphi/dt, SSE_bdry
  const double cN = Vol / Nglob, cNL = Vol / Nglob / (double)L;
  const long P = (long)N * L;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long)gridDim.x * blockDim.x) {
    const int l = (int)(p / N), n = (int)(p - (long)l * N);
    const double ul = u[p], vl = v[p];
    const double wl = w_per_point ? w[p] : w[n];
    const double phi = vl * wl;
    double phit = wl * vt[p];                         // d(phi)/dt = w dv/dt + v dw/dt
    if (wt != nullptr) phit += vl * wt[p];
    const double cl = c != nullptr ? c[p] : ckappa * ul;
    double s3 = cl * ul * phi + f[p] * phi;           // src/loss.py:70
    double I = 0.0;
    if (l == 0) {
      const double hn = h[n];
      double s31;
      s31 = s3x != nullptr ? s3x[n] : 0.0;            // (a = identity, b = 0: contracted below, spread over the time rows)
      s3 += s31;                                      // src/loss.py:66-69 (only non-zero at l = 0)
      I -= cN * hn * vl;                              // s1, src/loss.py:64
      const double hr = href != nullptr ? href[n] : hn;
      acc[2] += (ul - hr) * (ul - hr);                // src/loss.py:79
    }
    if (s3x == nullptr) {
      // a = identity, b = 0:  s31_n = sum_i d_i phi d_i u  with  nabla phi = w nabla v + v nabla w  (at t_0, on the
      // v-sample) and  nabla u = G_n = d(sum_l u)/dx_n + d(sum_l u)/d(start) nabla h   (SURVEY Appendix A Q3).
      // The term belongs to the l = 0 point of path n; the thread of (l, n) adds the dimensions i = l, l + L, ... so that
      // the d-long contraction is spread over all L time rows instead of loading the first N threads 20-fold.
      double part = 0.0;
      if (l < d) {
        const double w0n = w0[n], gsn = gs[n], v0n = v[n];
        for (int i = l; i < d; i += L) {
          const long q = (long)i * N + n;
          part += (w0n * gxv[q] + v0n * gwx0T[q]) * (gx[q] + gsn * ghT[q]);
        }
      }
      I += cNL * s3_scale * part;
    }
    if (l == L - 1) I += cN * ul * vl;
    if (pairwise) {
      I += cNL * s3_scale * s3;
      acc[3] += ul;
      acc[4] += phit;
    } else {
      I -= cNL * (ul * phit - s3_scale * s3);         // -(s2 - s3), src/loss.py:65,71-73
    }
    acc[0] += I;
    acc[1] += vl * vl;
  }
  // boundary penalty sum (u_b - g)^2 (src/loss.py:84) over the boundary paths of the group, when the caller hands them over: what
  // xw_bdry_partials does in a launch of its own -- one launch and one ticket round less per generator sub-step (4 % of the
  // sub-step of a 512-path shard, profiles/r04_what_if_shard512.txt)
  if (ub != nullptr)
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < Pb; p += (long)gridDim.x * blockDim.x) {
      const double r = ub[p] - gb[p];
      acc[5] += r * r;
    }
  const bool last = grid_sum<6>(acc, work, scal);
  if (finalize && last) {
    // the block that completed the sums also turns them into the loss values (what xw_losses does) -- one launch and
    // one dependency edge less per sub-step; scal[3] (boundary SSE) is complete too: summed above, or by an earlier launch
    __syncthreads();
    if (threadIdx.x == 0) {
      if (step != nullptr) *step += 1;
      if (pairwise) fold_pairs(scal, cN);
      loss_values(scal, L, Lb, Vol, Nglob, Nbglob, alpha, init_off, bdry_off);
    }
  }
}

__global__ void __launch_bounds__(1024) k_bdry(const double* __restrict__ ub, const double* __restrict__ gb, long P,
                                              double coef, double* __restrict__ ubar_b, double* __restrict__ work,
                                              double* __restrict__ scal) {
  double acc[1] = {0.0};
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long)gridDim.x * blockDim.x) {
    const double r = ub[p] - gb[p];
    acc[0] += r * r;                                  // src/loss.py:84
    if (ubar_b != nullptr) ubar_b[p] = coef * r;
  }
  grid_sum<1>(acc, work, scal + 3);
}


// Generator cotangent bases (both available right after the forward passes -- neither needs the global I):
//   ubarA = pollution + alpha * d(init)/du        ubarB = dI/du
// d loss_u/d theta = J^T ubarA + (2/I) J^T ubarB (+ the boundary sweep); the 2/I is applied inside the Adam kernel.
__global__ void __launch_bounds__(256) k_gen_cots(const double* __restrict__ u, const double* __restrict__ v,
                                                  const double* __restrict__ w, int w_per_point,
                                                  const double* __restrict__ c, const double* __restrict__ cp,
                                                  double ckappa, const double* __restrict__ h, int N, int L, double Vol,
                                                  double Nglob, double alpha, double pollution,
                                                  const double* __restrict__ scal, double* __restrict__ ubarA,
                                                  double* __restrict__ ubarB) {
  // scal != NULL: merged form -- the global I = scal[0] is known, write ubarA := A + (2/I) B (ubarB is not written)
  const double cI = scal != nullptr ? 2.0 / scal[0] : 0.0;
  const double cN = Vol / Nglob, cNL = Vol / Nglob / (double)L;
  const long P = (long)N * L;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long)gridDim.x * blockDim.x) {
    const int l = (int)(p / N), n = (int)(p - (long)l * N);
    const double ul = u[p];
    const bool needB = ubarB != nullptr || scal != nullptr;
    const double vl = needB ? v[p] : 0.0;              // basis A does not read the test network at all
    const double wl = w_per_point ? w[p] : w[n];
    const double dcu = c != nullptr ? c[p] + ul * cp[p] : 2.0 * ckappa * ul;   // d(c(u) u)/du
    double gB = 0.0;
    if (needB) {
      gB = cNL * dcu * vl * wl;                        // through c u phi  (src/loss.py:70)
      if (l == L - 1) gB += cN * vl;                   // through s1       (src/loss.py:64)
      if (ubarB != nullptr) ubarB[p] = gB;
    }
    if (ubarA != nullptr) {
      double gA = pollution;                           // helper backward  (src/loss.py:55)
      if (l == 0) gA += alpha * 2.0 * (ul - h[n]) / Nglob;   // initial penalty  (src/loss.py:79,93)
      ubarA[p] = gA + cI * gB;
    }
  }
}

__global__ void k_losses(double* __restrict__ scal, int L, int Lb, double Vol, double Nglob, double Nbglob, double alpha,
                         double init_off, double bdry_off, long long* __restrict__ step) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (step != nullptr) *step += 1;                  // optimiser step counter (after xw_adam has read it)
    loss_values(scal, L, Lb, Vol, Nglob, Nbglob, alpha, init_off, bdry_off);
  }
}
__global__ void k_pair_fold(double* __restrict__ scal, double cN) {
  if (blockIdx.x == 0 && threadIdx.x == 0) fold_pairs(scal, cN);
}

// Distance weight of the hypercube [bot, top]^d and its x-gradient at one point per thread (src/dataset.py:278-282 and the
// autograd pass through it, src/loss.py:51-63): w = min(min_i |top - x_i|, min_i |bot - x_i|) in the sample's float32 arithmetic,
// the gradient as autograd returns it -- min over a dimension sends it to the FIRST index that attains the minimum,
// torch.minimum splits it half / half on equal operands, |.| has slope sign(.) (0 at 0) -- both widened to float64 on the
// way out, into the layouts the weak form reads ([N], [d, N]); optionally the transposed float64 copy of the points.
__global__ void __launch_bounds__(256) k_cube_weight(const float* __restrict__ x, int N, int d, float top, float bot,
                                                     double* __restrict__ w, double* __restrict__ w0,
                                                     double* __restrict__ gwT, double* __restrict__ xT) {
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
    const float* xn = x + (long)n * d;
    float at = fabsf(top - xn[0]), ab = fabsf(bot - xn[0]);
    int it = 0, ib = 0;
    for (int i = 1; i < d; ++i) {
      const float a = fabsf(top - xn[i]), b = fabsf(bot - xn[i]);
      if (a < at) { at = a; it = i; }
      if (b < ab) { ab = b; ib = i; }
    }
    const float wv = ab < at ? ab : at;
    const float ct = (at < ab ? 1.0f : 0.0f) + (at == ab ? 0.5f : 0.0f), cb = (ab < at ? 1.0f : 0.0f) + (at == ab ? 0.5f : 0.0f);
    const float dt = top - xn[it], db = bot - xn[ib];
    const float gt = -(dt > 0.0f ? 1.0f : (dt < 0.0f ? -1.0f : 0.0f)) * ct, gb = -(db > 0.0f ? 1.0f : (db < 0.0f ? -1.0f : 0.0f)) * cb;
    w[n] = (double)wv;
    if (w0) w0[n] = (double)wv;
    for (int i = 0; i < d; ++i) {
      float g = 0.0f;
      if (i == it) g += gt;
      if (i == ib) g += gb;
      gwT[(long)i * N + n] = (double)g;
      if (xT) xT[(long)i * N + n] = (double)xn[i];
    }
  }
}

__global__ void __launch_bounds__(256) k_disc_cot(const double* __restrict__ u, const double* __restrict__ v,
                                                  const double* __restrict__ w, int w_per_point,
                                                  const double* __restrict__ c, double ckappa,
                                                  const double* __restrict__ f, const double* __restrict__ h, int N,
                                                  int L, double Vol, double Nglob, double pollution, double s3_scale,
                                                  const double* __restrict__ scal_in, double* __restrict__ vbar) {
  const double I = scal_in[0], S = scal_in[1];
  const double cI = 2.0 / I;
  const double cN = Vol / Nglob, cNL = Vol / Nglob / (double)L;
  const long P = (long)N * L;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long)gridDim.x * blockDim.x) {
    const int l = (int)(p / N), n = (int)(p - (long)l * N);
    const double ul = u[p], vl = v[p];
    const double wl = w_per_point ? w[p] : w[n];
    const double cl = c != nullptr ? c[p] : ckappa * ul;
    double dI = cNL * s3_scale * (cl * ul + f[p]) * wl;   // d I / d v through phi = v w in c u phi + f phi
    if (l == L - 1) dI += cN * ul;
    if (l == 0) dI -= cN * h[n];
    vbar[p] = pollution * wl - cI * dI + 2.0 * vl / S;  // loss_v = -(log I^2 - log(V S / P))
  }
}

// torch.optim.Adam (betas, eps defaults; no weight decay, no amsgrad) fused with the reduction of the per-wave gradient
// slabs:   g = eA + sum_s A[s] + coefB * (eB + sum_s B[s]),   coefB = scal ? 2 / scal[0] : 1
// (A: cotangent basis that needs no global scalar, B: the dI/du basis scaled by d log(I^2)/dI).
// Block = 16 parameters x 64 slab groups; grid = P / 16.
#define XW_ADAM_PARAMS 16
#define XW_ADAM_GROUPS 64
// (eA, eB and gsum_out carry no __restrict__: the group runner hands the SAME buffer in as the extra gradient and as the output
//  of the summed gradient -- xw_substep.hip, a sharded discriminator update -- and every thread reads its entry before it writes it)
__global__ void __launch_bounds__(1024) k_adam(double* __restrict__ param, const double* __restrict__ gA, int nA,
                                               const double* eA, const double* __restrict__ gB, int nB,
                                               const double* eB, const double* __restrict__ scal,
                                               double* __restrict__ m, double* __restrict__ v,
                                               const long long* __restrict__ step, int step_is_current, int P, double lr,
                                               double beta1, double beta2, double eps, double* gsum_out,
                                               int lag_lo, int lag_hi, int skip, const long long* __restrict__ lag) {
  // block = 16 parameters (one 128-byte line per slab row) x 64 slab groups: P / 16 blocks spread the 10 MB of slabs of
  // a generator sub-step over ~100 CUs (64 parameters per block used 26 of them and took 17 us)
  __shared__ double red[2][XW_ADAM_GROUPS][XW_ADAM_PARAMS];
  const int tx = threadIdx.x % XW_ADAM_PARAMS, ty = threadIdx.x / XW_ADAM_PARAMS;
  const int i = blockIdx.x * XW_ADAM_PARAMS + tx;
  double a = 0.0, b = 0.0;
  if (i < P) {
    for (int s = ty; s < nA; s += XW_ADAM_GROUPS) a += gA[(long)s * P + i];
    for (int s = ty; s < nB; s += XW_ADAM_GROUPS) b += gB[(long)s * P + i];
  }
  red[0][ty][tx] = a;
  red[1][ty][tx] = b;
  __syncthreads();
  if (ty != 0 || i >= P) return;
  a = eA != nullptr ? eA[i] : 0.0;
  b = eB != nullptr ? eB[i] : 0.0;
#pragma unroll 8
  for (int k = 0; k < XW_ADAM_GROUPS; ++k) {
    a += red[0][k][tx];
    b += red[1][k][tx];
  }
  const double coefB = scal != nullptr ? 2.0 / scal[0] : 1.0;
  const double g = a + coefB * b;
  if (gsum_out != nullptr) gsum_out[i] = g;
  // parameters [lag_lo, lag_hi) (the field of u_theta) have their own step count, *step - *lag: torch's Adam skips a
  // parameter whose .grad is None -- no moment decay, no step -- which is what the field's are while no group of the
  // sub-iteration has integrated the ODE (single-slice groups, src/model.py:89-91; zero_grad() of torch >= 2.0)
  const bool lagged = i >= lag_lo && i < lag_hi;
  if (lagged && skip) return;
  const long long t = *step + (step_is_current ? 0 : 1) - (lagged && lag != nullptr ? *lag : 0);
  const double bc1 = 1.0 - pow(beta1, (double)t), bc2 = 1.0 - pow(beta2, (double)t);
  const double mi = beta1 * m[i] + (1.0 - beta1) * g;
  const double vi = beta2 * v[i] + (1.0 - beta2) * g * g;
  m[i] = mi;
  v[i] = vi;
  param[i] -= (lr / bc1) * (mi / (sqrt(vi) / sqrt(bc2) + eps));
}
__global__ void k_step_inc(long long* step) { *step += 1; }

// sum of the per-wave gradient slabs (several GPUs: into the exchange buffer).  Same shape as the reduction inside k_adam:
// block = 16 parameters x 64 slab groups, P / 16 blocks, a fixed summation tree.  (One thread per parameter walking all
// slabs -- 7 blocks for 1551 parameters -- took 120 us for the 512 slabs of a sweep launch: three such launches per
// sub-step were a third of a multi-GPU sub-step.)
__global__ void __launch_bounds__(1024) k_slab_sum(const double* __restrict__ gslab, int nslab, int P, int accumulate,
                                                   double* __restrict__ out) {
  __shared__ double red[XW_ADAM_GROUPS][XW_ADAM_PARAMS];
  const int tx = threadIdx.x % XW_ADAM_PARAMS, ty = threadIdx.x / XW_ADAM_PARAMS;
  const int i = blockIdx.x * XW_ADAM_PARAMS + tx;
  double a = 0.0;
  if (i < P)
    for (int s = ty; s < nslab; s += XW_ADAM_GROUPS) a += gslab[(long)s * P + i];
  red[ty][tx] = a;
  __syncthreads();
  if (ty != 0 || i >= P) return;
  double g = accumulate ? out[i] : 0.0;
#pragma unroll 8
  for (int k = 0; k < XW_ADAM_GROUPS; ++k) g += red[k][tx];
  out[i] = g;
}

// Launch shape of the deterministic grid sums (k_weak_partials, k_bdry).  Every block ends with one atomic on ONE ticket word, and
// the blocks of such a launch finish together: 512 blocks of 256 threads queued for ~6 us on that word (one word serves ~88
// atomics per us) in a kernel that is on the critical path of every sub-step.  128 blocks of 1024 threads: 17.2 -> 11.3 us
// for the 131,072 points of the headline group, 0.4794 -> 0.4717 ms per sub-step (tools/cap_sweep.sh; 256 x 512: 12.4 us, 64 x 1024: 12.5,
// 32 x 1024: 15).  XW_REDUCE_BLOCKS / XW_REDUCE_THREADS override (measurements).
inline int reduce_cap(long points) {
  static const int v = [] { const char* e = getenv("XW_REDUCE_BLOCKS"); const int x = e ? atoi(e) : 0; return x > 0 && x <= 1024 ? x : 0; }();
  // (a million points and more -- BASELINE configs[2], [3] whole on one GPU -- stream through 256 blocks faster: 62 against 85 us at 2 M)
  return v ? v : (points > (1L << 18) ? 256 : 128);
}
inline int reduce_threads() {
  static const int v = [] { const char* e = getenv("XW_REDUCE_THREADS"); const int x = e ? atoi(e) : 0; return (x == 256 || x == 512) ? x : 1024; }();
  return v;
}
inline int blocks_for(long n, int per, int cap) {
  long b = (n + per - 1) / per;
  if (b < 1) b = 1;
  return (int)(b > cap ? cap : b);
}


// s3x[n] = sum_ij a_ij d_i(phi) d_j(u) + phi sum_i b_i d_i(u)  at the first time index (src/loss.py:66-69; nabla u only
// exists there), with  nabla phi = w nabla v + v nabla w  and  nabla u = gx + gs nabla h.
// a is given in the cheapest form the caller has: amode 0 identity, 1 one matrix A0[d,d] for all points, 2 a diagonal
// A0[d,N], 3 the full table A0[d,d,N] (row-major over (i, j), paths fastest: consecutive lanes = consecutive paths).
// One lane per path, d^2 coalesced loads of a: HBM-streaming (8 d^2 N bytes: 655 MB at d = 100, N = 8192).
__global__ void __launch_bounds__(256) k_weak_contract(const double* __restrict__ A0, int amode,
                                                       const double* __restrict__ B0, const double* __restrict__ gx,
                                                       const double* __restrict__ gs, const double* __restrict__ ghT,
                                                       const double* __restrict__ gxv, const double* __restrict__ w0,
                                                       const double* __restrict__ gwx0T, const double* __restrict__ v0,
                                                       int d, int N, double* __restrict__ s3x) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const double w = w0[n], v = v0[n], gsn = gs[n];
  double acc = 0.0, bsum = 0.0;
  for (int i = 0; i < d; ++i) {
    const double dphi = w * gxv[(long)i * N + n] + v * gwx0T[(long)i * N + n];
    const double dui = gx[(long)i * N + n] + gsn * ghT[(long)i * N + n];
    if (B0 != nullptr) bsum += B0[(long)i * N + n] * dui;
    if (amode == 0) {
      acc += dphi * dui;
    } else if (amode == 2) {
      acc += A0[(long)i * N + n] * dphi * dui;
    } else {
      double row = 0.0;
      for (int j = 0; j < d; ++j) {
        const double duj = gx[(long)j * N + n] + gsn * ghT[(long)j * N + n];
        const double a = amode == 1 ? A0[i * d + j] : A0[((long)i * d + j) * N + n];
        row += a * duj;
      }
      acc += dphi * row;
    }
  }
  s3x[n] = acc + v * w * bsum;
}
}  // namespace

extern "C" int xw_weak_partials(const double* u, const double* v, const double* vt, const double* w, int w_per_point,
                                const double* wt, const double* s3x, const double* gx, const double* gs, const double* ghT,
                                const double* gxv, const double* w0, const double* gwx0T, int d, const double* c,
                                double ckappa, const double* f, const double* h, const double* href, int pairwise,
                                double s3_scale, int N, int L, double Vol, double Nglob, double* work, double* scal,
                                int finalize, int Lb, double Nbglob, double alpha, double init_off, double bdry_off,
                                long long* step, const double* ub, const double* gb, long Pb, void* stream) {
  if (!u || !v || !vt || !w || !f || !h || !work || !scal || N <= 0 || L <= 0 || (finalize && Lb <= 0)) return XW_E_ARG;
  if ((ub == nullptr) != (gb == nullptr) || (ub != nullptr && Pb <= 0)) return XW_E_ARG;
  if (!s3x && (!gx || !gs || !ghT || !gxv || !w0 || !gwx0T || d <= 0)) return XW_E_ARG;
  if (pairwise && L != 1) return XW_E_ARG;
  hipLaunchKernelGGL(k_weak_partials, dim3(blocks_for((long)N * L, reduce_threads(), reduce_cap((long)N * L))), dim3(reduce_threads()), 0, (hipStream_t)stream, u, v, vt, w,
                     w_per_point, wt, s3x, gx, gs, ghT, gxv, w0, gwx0T, d, c, ckappa, f, h, href, pairwise, s3_scale, N, L, Vol,
                     Nglob, work, scal, finalize, Lb, Nbglob, alpha, init_off, bdry_off, step, ub, gb, Pb);
  return xw_launch_status();
}

extern "C" int xw_bdry_partials(const double* ub, const double* g, int Nb, int L, double alpha, double Nbglob,
                                double* ubar_b, double* work, double* scal, void* stream) {
  if (!ub || !g || !work || !scal || Nb <= 0 || L <= 0) return XW_E_ARG;
  const long P = (long)Nb * L;
  const double coef = alpha * 2.0 / (Nbglob * (double)L);
  hipLaunchKernelGGL(k_bdry, dim3(blocks_for(P, reduce_threads(), reduce_cap(P))), dim3(reduce_threads()), 0, (hipStream_t)stream, ub, g, P, coef, ubar_b, work, scal);
  return xw_launch_status();
}

extern "C" int xw_gen_cotangents(const double* u, const double* v, const double* w, int w_per_point, const double* c,
                                 const double* cp, double ckappa, const double* h, int N, int L, double Vol, double Nglob,
                                 double alpha, double pollution, const double* scal, double* ubarA, double* ubarB,
                                 void* stream) {
  if (!u || !w || !h || (!ubarA && !ubarB) || ((ubarB || scal) && !v) || (scal && (!ubarA || ubarB)) || N <= 0 || L <= 0)
    return XW_E_ARG;
  if ((c == nullptr) != (cp == nullptr)) return XW_E_ARG;
  hipLaunchKernelGGL(k_gen_cots, dim3(blocks_for((long)N * L, 256, 2048)), dim3(256), 0, (hipStream_t)stream, u, v, w,
                     w_per_point, c, cp, ckappa, h, N, L, Vol, Nglob, alpha, pollution, scal, ubarA, ubarB);
  return xw_launch_status();
}

extern "C" int xw_losses(double* scal, int L, int Lb, double Vol, double Nglob, double Nbglob, double alpha,
                         double init_off, double bdry_off, long long* step, void* stream) {
  if (!scal || L <= 0 || Lb <= 0) return XW_E_ARG;
  hipLaunchKernelGGL(k_losses, dim3(1), dim3(64), 0, (hipStream_t)stream, scal, L, Lb, Vol, Nglob, Nbglob, alpha, init_off,
                     bdry_off, step);
  return xw_launch_status();
}

extern "C" int xw_pair_fold(double* scal, double Vol, double Nglob, void* stream) {
  if (!scal || Nglob <= 0) return XW_E_ARG;
  hipLaunchKernelGGL(k_pair_fold, dim3(1), dim3(64), 0, (hipStream_t)stream, scal, Vol / Nglob);
  return xw_launch_status();
}

extern "C" int xw_cube_weight(const float* x, int N, int d, double top, double bot, double* w, double* w0, double* gwT,
                              double* xT, void* stream) {
  if (!x || !w || !gwT || N <= 0 || d <= 0 || !(top > bot)) return XW_E_ARG;
  hipLaunchKernelGGL(k_cube_weight, dim3(blocks_for(N, 256, 1024)), dim3(256), 0, (hipStream_t)stream, x, N, d, (float)top,
                     (float)bot, w, w0, gwT, xT);
  return xw_launch_status();
}

// ---- sample fields of the groups of a list-domain sample in one launch ------------------------------------------------------
// A sample of a time-varying ball domain is 11-20 groups; per group the engine needs ~18 device arrays that are all STRIDED
// VIEWS of what the sampler uploaded and the callables returned for the whole sample (the time grid, transposed coordinates,
// [N, L] tables as [L, N], one component of a gradient, ...).  Taken one by one through tensor operations that was ~45 host
// operations per group -- the largest item of an outer iteration's host time once nothing else waited.  Here
// every output array is one row of a table {src, dst, n0 x n1 x n2, strides of src}: dst[i][j][k] = src[i s0 + j s1 + k s2],
// dst contiguous; the table of the whole sample (~350 rows) is uploaded once and ONE launch walks all elements.
namespace {
__global__ __launch_bounds__(256) void k_gather_fields(const XwGather* __restrict__ tab, int count, long total) {
  __shared__ long first[1025];                    // first[r] = elements before row r; first[count] = total
  for (int r = threadIdx.x; r <= count; r += blockDim.x) first[r] = r < count ? tab[r].before : total;
  __syncthreads();
  const long stride = (long)gridDim.x * blockDim.x;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    int lo = 0, hi = count;                       // the row with first[lo] <= e < first[lo + 1]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (first[mid] <= e) lo = mid; else hi = mid;
    }
    const XwGather g = tab[lo];
    const long local = e - g.before;
    const long k = local % g.n2, ij = local / g.n2, j = ij % g.n1, i = ij / g.n1;
    g.dst[local] = g.src[i * g.s0 + j * g.s1 + k * g.s2];
  }
}
}  // namespace

extern "C" int xw_gather_fields(const XwGather* table_dev, int count, long total, void* stream) {
  if (!table_dev || count <= 0 || count > 1024 || total <= 0) return XW_E_ARG;
  hipLaunchKernelGGL(k_gather_fields, dim3(blocks_for(total, 256, 2048)), dim3(256), 0, (hipStream_t)stream, table_dev, count, total);
  return xw_launch_status();
}

extern "C" int xw_disc_cotangent(const double* u, const double* v, const double* w, int w_per_point, const double* c,
                                 double ckappa, const double* f, const double* h, int N, int L, double Vol, double Nglob,
                                 double pollution, double s3_scale, const double* scal_in, double* vbar, void* stream) {
  if (!u || !v || !w || !f || !h || !scal_in || !vbar || N <= 0 || L <= 0) return XW_E_ARG;
  hipLaunchKernelGGL(k_disc_cot, dim3(blocks_for((long)N * L, 256, 2048)), dim3(256), 0, (hipStream_t)stream, u, v, w,
                     w_per_point, c, ckappa, f, h, N, L, Vol, Nglob, pollution, s3_scale, scal_in, vbar);
  return xw_launch_status();
}

extern "C" int xw_adam(double* param, const double* gslabA, int nA, const double* gextraA, const double* gslabB, int nB,
                       const double* gextraB, const double* scal, double* m, double* v, long long* step, int bump_step,
                       int P, double lr, double beta1, double beta2, double eps, double* gsum_out, int lag_lo, int lag_hi,
                       int skip, long long* lag, void* stream) {
  // bump_step: 1 = this call advances the counter after the update; 0 = the counter was left alone (caller advances it
  // later);  -1 = the counter was ALREADY advanced for this update (xw_losses ran first), use it as is
  // lag_lo..lag_hi: parameter range with its own step count *step - *lag (NULL / empty range: none); skip != 0: that
  // range is left untouched by this update and *lag advances (torch's Adam skipping parameters without a gradient)
  if (!param || !m || !v || !step || P <= 0 || nA < 0 || nB < 0 || (nA > 0 && !gslabA) || (nB > 0 && !gslabB)) return XW_E_ARG;
  if (lag_lo < 0 || lag_hi > P || (skip && (!lag || lag_hi <= lag_lo))) return XW_E_ARG;
  hipLaunchKernelGGL(k_adam, dim3((P + XW_ADAM_PARAMS - 1) / XW_ADAM_PARAMS), dim3(1024), 0, (hipStream_t)stream, param, gslabA, nA, gextraA, gslabB,
                     nB, gextraB, scal, m, v, step, bump_step < 0 ? 1 : 0, P, lr, beta1, beta2, eps, gsum_out, lag_lo, lag_hi, skip,
                     lag);
  if (skip) hipLaunchKernelGGL(k_step_inc, dim3(1), dim3(1), 0, (hipStream_t)stream, lag);
  if (bump_step > 0) hipLaunchKernelGGL(k_step_inc, dim3(1), dim3(1), 0, (hipStream_t)stream, step);
  return xw_launch_status();
}

// two slab sets in one launch (blockIdx.y picks the set): the generator sub-step's [sum A | sum B] exchange buffer
namespace {
__global__ void __launch_bounds__(1024) k_slab_sum2(const double* __restrict__ gA, int nA, double* __restrict__ outA,
                                                    const double* __restrict__ gB, int nB, double* __restrict__ outB, int P) {
  __shared__ double red[XW_ADAM_GROUPS][XW_ADAM_PARAMS];
  const double* __restrict__ gslab = blockIdx.y == 0 ? gA : gB;
  const int nslab = blockIdx.y == 0 ? nA : nB;
  double* __restrict__ out = blockIdx.y == 0 ? outA : outB;
  const int tx = threadIdx.x % XW_ADAM_PARAMS, ty = threadIdx.x / XW_ADAM_PARAMS;
  const int i = blockIdx.x * XW_ADAM_PARAMS + tx;
  double a = 0.0;
  if (i < P)
    for (int s = ty; s < nslab; s += XW_ADAM_GROUPS) a += gslab[(long)s * P + i];
  red[ty][tx] = a;
  __syncthreads();
  if (ty != 0 || i >= P) return;
  double g = 0.0;
#pragma unroll 8
  for (int k = 0; k < XW_ADAM_GROUPS; ++k) g += red[k][tx];
  out[i] = g;
}
}  // namespace
extern "C" int xw_slab_sum2(const double* gA, int nA, double* outA, const double* gB, int nB, double* outB, int P, void* stream) {
  if (!gA || !gB || !outA || !outB || P <= 0 || nA <= 0 || nB <= 0) return XW_E_ARG;
  hipLaunchKernelGGL(k_slab_sum2, dim3((P + XW_ADAM_PARAMS - 1) / XW_ADAM_PARAMS, 2), dim3(XW_ADAM_PARAMS * XW_ADAM_GROUPS), 0,
                     (hipStream_t)stream, gA, nA, outA, gB, nB, outB, P);
  return xw_launch_status();
}

extern "C" int xw_slab_sum(const double* gslab, int nslab, int P, int accumulate, double* out, void* stream) {
  if (!gslab || !out || P <= 0 || nslab <= 0) return XW_E_ARG;
  hipLaunchKernelGGL(k_slab_sum, dim3((P + XW_ADAM_PARAMS - 1) / XW_ADAM_PARAMS), dim3(XW_ADAM_PARAMS * XW_ADAM_GROUPS), 0,
                     (hipStream_t)stream, gslab, nslab, P, accumulate, out);
  return xw_launch_status();
}

extern "C" int xw_abi_version(void) { return 31; }
extern "C" int xw_reduce_work_size(void) { return 6 * 1024 + 8; }

extern "C" int xw_supported_dims(char* buf, int buflen) {
  static const char s[] = "ode (H,K)=(20,10),(32,12),(64,16), m=1..10 [MFMA]; any other H<=64, K<=16, m<=32 [generic path: vector ALU, slow]; disc_fwd W=50,64,96,128 any q; disc_bwd W=50 (q=9 unrolled, any q from the record), W=64,96,128 (from the record) [MFMA]; any other W<=128, q<=16 [generic path, from the record]; d<=126";
  int i = 0;
  for (; s[i] && i < buflen - 1; ++i) buf[i] = s[i];
  if (buflen > 0) buf[i] = 0;
  return i;
}

extern "C" int xw_theta_size(int d, int H, int K) { return u_offsets(d, H, K).total; }
extern "C" int xw_phi_size(int d, int W) { return v_offsets(d, W).total; }

extern "C" int xw_weak_contract_general(const double* A0, int amode, const double* B0, const double* gx, const double* gs,
                                        const double* ghT, const double* gxv, const double* w0, const double* gwx0T,
                                        const double* v0, int d, int N, double* s3x, void* stream) {
  if (!gx || !gs || !ghT || !gxv || !w0 || !gwx0T || !v0 || !s3x || d <= 0 || N <= 0 || amode < 0 || amode > 3 ||
      (amode != 0 && !A0))
    return XW_E_ARG;
  hipLaunchKernelGGL(k_weak_contract, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, A0, amode, B0, gx, gs, ghT, gxv,
                     w0, gwx0T, v0, d, N, s3x);
  return xw_launch_status();
}
