// xw_substep.hip -- one optimiser sub-step of one group of paths as ONE call (include/xnwan.h: xw_substep_gen / _disc).
//
// Host code only: the chain of launches that engine.py issues for a group of a list domain (the reference's
// `for (datau, datav, bdata) in points:` body, src/training.py:127-138,152-162), in the same order with the same arguments,
// on one stream.  A cone sample has 11-12 groups, an hourglass sample 20, each with its own shapes every sample: no graph to
// replay, and ~15 launches per group and sub-step issued from Python cost more host time than the GPU needs to run them.
#include <hip/hip_runtime.h>
#include <string.h>
#include "xnwan.h"

namespace {
// Two independent chains of a sub-step -- the test network and the stepper's forward pass + sweeps -- run on the caller's
// stream and on a side stream of the library (one per device, created at first use; forks and joins are event record / wait
// pairs, ~1 us each from C where the Python stream contexts cost more than the overlap returned): a group of a few hundred
// paths is launch- and latency-bound, and its critical path shrinks from the sum of the two chains to the longer one.
struct Side {
  hipStream_t s = nullptr;
  hipEvent_t fork = nullptr, fwd = nullptr, done = nullptr;
};
Side* side_of_current_device() {
  static Side sides[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  Side& x = sides[dev];
  if (x.s == nullptr) {
    if (hipStreamCreateWithFlags(&x.s, hipStreamNonBlocking) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&x.fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&x.fwd, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&x.done, hipEventDisableTiming) != hipSuccess)
      return nullptr;
  }
  return &x;
}
#define XW_HIP(call) { const hipError_t e_ = (call); if (e_ != hipSuccess) return (int)e_; }

inline XwOdeFwdJob fwd_job(const double* xT, const double* start, double* u, double* Y, double* act, int N, int x_only, int narrow) {
  XwOdeFwdJob j;
  j.xT = xT; j.start = start; j.u = u; j.Y = Y; j.act = act; j.N = N; j.act_x_only = x_only; j.narrow = narrow; j.prio_drop = 0;
  return j;
}
inline XwOdeBwdJob bwd_job(const double* xT, const double* start, const double* Y, const double* act, int N) {
  XwOdeBwdJob j;
  memset(&j, 0, sizeof(j));
  j.xT = xT; j.start = start; j.Y = Y; j.act = act; j.N = N;
  return j;
}
#define XW_TRY(call) { const int rc_ = (call); if (rc_ != 0) return rc_; }

// v, dv/dt at all points and nabla_x v at the first time index (path mode, or point mode when the paths of the group do not
// share a time column); optionally the record of layer inputs
int test_net(const XwGroup* g, const XwSolverState* s, int blocks, double* record, void* stream) {
  if (g->tpp != nullptr)
    return xw_disc_fwd(g->xvT_pts, nullptr, g->tpp, s->phi, g->N * g->L, 1, g->d, s->W, s->q, g->v, g->vt, g->gxv, g->gtv, g->N, blocks,
                       record, stream);
  if (g->xproj != nullptr) XW_TRY(xw_disc_xproj(g->xvT, s->phi, g->N, g->d, s->W, g->xproj, stream));
  return xw_disc_fwd_xproj(g->xvT, g->t, nullptr, s->phi, g->N, g->L, g->d, s->W, s->q, g->v, g->vt, g->gxv, g->gtv, g->N, blocks, record,
                           g->xproj, stream);
}
// I, sum v^2, SSE_init (+ finalize: loss values and the optimiser's counter -- one process, the sums are global)
int contract(const XwGroup* g, const XwSolverState* s, long long* step, bool with_bdry, bool finalize, void* stream) {
  const double* s3x = nullptr;
  if (g->A0 != nullptr || g->B0 != nullptr) {
    XW_TRY(xw_weak_contract_general(g->A0, g->amode, g->B0, g->gx, g->gs, g->ghT, g->gxv, g->w0, g->gwx0T, g->v, g->d, g->N, g->s3x,
                                    stream));
    s3x = g->s3x;
  }
  const bool fused = s3x == nullptr;
  return xw_weak_partials(g->u, g->v, g->vt, g->w, g->w_per_point, g->wt, s3x, fused ? g->gx : nullptr, fused ? g->gs : nullptr,
                          fused ? g->ghT : nullptr, fused ? g->gxv : nullptr, fused ? g->w0 : nullptr, fused ? g->gwx0T : nullptr,
                          fused ? g->d : 0, g->c, g->ckappa, g->f, g->h, g->pair_i ? g->href : nullptr, g->pair_i ? 1 : 0,
                          g->pair_i ? g->s3_scale : 1.0, g->N, g->L, g->Vol, g->Nglob, g->work_i, s->scal, finalize ? 1 : 0,
                          g->Lb > 0 ? g->Lb : 1, g->Nbglob, s->alpha, g->init_off, g->bdry_off, finalize ? step : nullptr, with_bdry && g->Nb > 0 ? g->ub : nullptr,
                          with_bdry && g->Nb > 0 ? g->g : nullptr, with_bdry && g->Nb > 0 ? (long)g->Nb * g->Lb : 0, stream);
}
inline bool sharded(const XwGroup* g, const XwSolverState* s) { return g->sharded != 0 && s->exchange != nullptr; }
#define XW_ZERO(ptr, count) XW_HIP(hipMemsetAsync((ptr), 0, sizeof(double) * (size_t)(count), (hipStream_t)stream))
}  // namespace

extern "C" int xw_substep_gen(const XwGroup* g, const XwSolverState* s, int skip_v, int store_record, double* accum,
                              int adam_skip_field, void* stream) {
  if (!g || !s || g->N < 0 || g->Nb < 0 || g->L <= 0) return XW_E_ARG;
  const bool shard = sharded(g, s);
  if (g->N == 0 && !shard) return XW_E_ARG;                 // (an empty interior share only exists on a sharded group)
  if (shard && (!s->pack_u || s->scal != s->pack_u + 2 * (long)s->Pu)) return XW_E_ARG;
  const bool fused_x = s->pollution == 1.0 && !s->adjoint;
  const bool have_i = g->N > 0, have_b = g->Nb > 0;
  const bool joint = have_i && have_b && g->same_grid;
  const int adj = s->adjoint ? 8 : 0;
  void* const main_stream = stream;
  if (have_i) {
    Side* sd = side_of_current_device();
    if (sd == nullptr) return XW_E_ARG;
    XW_HIP(hipEventRecord(sd->fork, (hipStream_t)main_stream));
    XW_HIP(hipStreamWaitEvent(sd->s, sd->fork, 0));
    if (!skip_v) XW_TRY(test_net(g, s, s->v_blocks, store_record ? g->vact : nullptr, main_stream));
    stream = (void*)sd->s;             // ---- side chain: forward pass, boundary residual, sweeps A (+ boundary)
    // u-forward: interior (+ boundary on the same grid) in one launch; the launch also clears the partial-sum slots
    {
      XwOdeFwdJob jobs[2] = {fwd_job(g->xT, g->start, g->u, g->Y, g->act, g->N, 0, (g->narrow >> 0) & 1),
                             fwd_job(g->xbT, g->start_b, g->ub, g->Yb, g->act_b, g->Nb, 0, (g->narrow >> 0) & 1)};
      XW_TRY(xw_ode_fwd_multi(jobs, joint ? 2 : 1, g->t, s->theta, s->method, g->L, g->d, s->H, s->K, s->m, s->scal, stream));
      if (have_b && !joint) {
        XwOdeFwdJob jb = fwd_job(g->xbT, g->start_b, g->ub, g->Yb, g->act_b, g->Nb, 0, (g->narrow >> 1) & 1);
        XW_TRY(xw_ode_fwd_multi(&jb, 1, g->tb, s->theta, s->method, g->Lb, g->d, s->H, s->K, s->m, nullptr, stream));
      }
    }
    XW_HIP(hipEventRecord(sd->fwd, sd->s));
    // ---- main chain again, behind the test network AND the forward pass: every sweep of the sub-step in ONE launch (three jobs),
    // the reduction and the update behind it on the same stream.  The groups this runner serves are small (a few tiles): their
    // sub-step is a chain of dependent launches, and each cross-queue dependency edge costs ~12 us -- this order has one (the
    // forward pass), the wide schedule of engine.py three (Engine._gen_front_compact is the same order for captured groups).
    stream = main_stream;
    XW_HIP(hipStreamWaitEvent((hipStream_t)main_stream, sd->fwd, 0));
  } else {
    // an empty interior share: nothing of the weak form lives here; the boundary paths this rank holds (if any) still take their
    // forward pass and sweep.  The partial-sum slots are cleared by hand (the interior forward pass is what clears them otherwise)
    XW_ZERO(s->scal, 16);
    if (have_b) {
      XwOdeFwdJob jb = fwd_job(g->xbT, g->start_b, g->ub, g->Yb, g->act_b, g->Nb, 0, (g->narrow >> 1) & 1);
      XW_TRY(xw_ode_fwd_multi(&jb, 1, g->tb, s->theta, s->method, g->Lb, g->d, s->H, s->K, s->m, nullptr, stream));
    }
  }
  // (the boundary sum of squares, a loss value only, is formed by the reduction at the end: contract(..., with_bdry))
  if (have_i && !fused_x) {     // the helper backward u.backward(ones) as a sweep of its own
    XwOdeBwdJob jx = bwd_job(g->xT, g->start, g->Y, g->act, g->N);
    jx.gx = g->gx; jx.gs = g->gs;
    XW_TRY(xw_ode_bwd_multi(&jx, 1, g->t, s->theta, s->method, g->L, g->d, s->H, s->K, s->m, 1 | adj | (((g->narrow >> 5) & 1) ? 16 : 0),
                            stream));
  }
  {
    // cotangent A (pollution + initial penalty, formed from the residual u - h at t_0), the boundary penalty, cotangent B = dI/du
    // (formed inside the sweep from u, v, w and c, c')
    XwOdeBwdJob jobs[3];
    int nj = 0;
    if (have_i) {
      jobs[nj] = bwd_job(g->xT, g->start, g->Y, g->act, g->N);
      jobs[nj].gslab = g->slabA;
      if (fused_x) { jobs[nj].gx = g->gx; jobs[nj].gs = g->gs; }
      jobs[nj].res_u = g->u; jobs[nj].res_ref = g->pair_i ? g->href : g->h; jobs[nj].res_first_only = 1;
      jobs[nj].res_coef = 2.0 * s->alpha / g->Nglob; jobs[nj].res_base = s->pollution;
      ++nj;
    }
    XwOdeBwdJob jb = bwd_job(g->xbT, g->start_b, g->Yb, g->act_b, g->Nb);
    if (have_b) {
      jb.gslab = g->slabA + (long)g->ns_u * s->Pu;
      jb.res_u = g->ub; jb.res_ref = g->g; jb.res_first_only = 0;
      jb.res_coef = 2.0 * s->alpha / (g->Nbglob * g->Lb); jb.res_base = 0.0;
    }
    if (joint) jobs[nj++] = jb;
    if (have_i) {
      XwOdeBwdJob jB = bwd_job(g->xT, g->start, g->Y, g->act, g->N);
      jB.gslab = g->slabB;
      jB.res_first_only = 2; jB.res_u = g->u; jB.res_ref = g->v;
      jB.res_coef = g->Vol / g->Nglob / g->L * g->s3_scale; jB.res_base = g->Vol / g->Nglob;
      jB.res_w_per_point = g->w_per_point; jB.res_w = g->w; jB.res_c = g->c; jB.res_cp = g->cp; jB.res_kappa2 = 2.0 * g->ckappa;
      jobs[nj++] = jB;
      const int modeA = (fused_x ? (1 | 2 | 4) : 2) | adj | (((g->narrow >> 2) & 1) ? 16 : 0);
      XW_TRY(xw_ode_bwd_multi(jobs, nj, g->t, s->theta, s->method, g->L, g->d, s->H, s->K, s->m, modeA, stream));
    }
    if (have_b && !joint)
      XW_TRY(xw_ode_bwd_multi(&jb, 1, g->tb, s->theta, s->method, g->Lb, g->d, s->H, s->K, s->m, 2 | adj | (((g->narrow >> 3) & 1) ? 16 : 0),
                              stream));
  }
  if (have_i) {
    XW_TRY(contract(g, s, s->step_u, true, !shard, stream));
  } else if (have_b) {
    XW_TRY(xw_bdry_partials(g->ub, g->g, g->Nb, g->Lb, s->alpha, g->Nbglob, nullptr, g->work_b, s->scal, stream));
  }
  if (!shard) {
    XW_TRY(xw_adam(s->theta, g->slabA, g->ns_u + g->ns_b, accum, g->slabB, g->ns_u, nullptr, s->scal, s->m_u, s->v_u, s->step_u, -1, s->Pu,
                   s->lr_u, s->beta1, s->beta2, s->eps, s->grad_u, s->lag_lo, s->lag_hi, adam_skip_field, s->lag_u, stream));
  } else {
    // ---- the ONE exchange of the generator sub-step: [sum of the A slabs | sum of the B slabs | partial sums] over the ranks
    const int nA = (have_i ? g->ns_u : 0) + (have_b ? g->ns_b : 0), nB = have_i ? g->ns_u : 0;
    double* const pA = s->pack_u;
    double* const pB = s->pack_u + s->Pu;
    if (nA > 0 && nB > 0) {
      XW_TRY(xw_slab_sum2(g->slabA, nA, pA, g->slabB, nB, pB, s->Pu, stream));
    } else {
      if (nA > 0) { XW_TRY(xw_slab_sum(g->slabA + (have_i ? 0 : (long)g->ns_u * s->Pu), nA, s->Pu, 0, pA, stream)); } else { XW_ZERO(pA, s->Pu); }
      XW_ZERO(pB, s->Pu);
    }
    XW_TRY(s->exchange(s->pack_u, 2 * s->Pu + 16, s->exchange_ctx, stream));
    if (g->pair_i) XW_TRY(xw_pair_fold(s->scal, g->Vol, g->Nglob, stream));
    XW_TRY(xw_losses(s->scal, g->L, g->Lb > 0 ? g->Lb : 1, g->Vol, g->Nglob, g->Nbglob, s->alpha, g->init_off, g->bdry_off, s->step_u, stream));
    // (the carried gradient of the sub-iteration's earlier groups rides in as a one-row slab set: global already)
    XW_TRY(xw_adam(s->theta, accum, accum != nullptr ? 1 : 0, pA, nullptr, 0, pB, s->scal, s->m_u, s->v_u, s->step_u, -1, s->Pu, s->lr_u,
                   s->beta1, s->beta2, s->eps, s->grad_u, s->lag_lo, s->lag_hi, adam_skip_field, s->lag_u, stream));
  }
  if (accum != nullptr) {
    const hipError_t e = hipMemcpyAsync(accum, s->grad_u, sizeof(double) * s->Pu, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}

extern "C" int xw_substep_disc(const XwGroup* g, const XwSolverState* s, int skip_v, int use_record, double* accum, void* stream) {
  if (!g || !s || g->N < 0 || g->L <= 0) return XW_E_ARG;
  const bool shard = sharded(g, s);
  if (g->N == 0 && !shard) return XW_E_ARG;
  const bool have_i = g->N > 0;
  const int adj = s->adjoint ? 8 : 0;
  double* record = use_record ? g->vact : nullptr;
  void* const main_stream = stream;
  if (have_i) {
    Side* sd = side_of_current_device();
    if (sd == nullptr) return XW_E_ARG;
    XW_HIP(hipEventRecord(sd->fork, (hipStream_t)main_stream));
    XW_HIP(hipStreamWaitEvent(sd->s, sd->fork, 0));
    if (!skip_v) XW_TRY(test_net(g, s, s->v_blocks_disc, record, main_stream));
    stream = (void*)sd->s;             // ---- side chain: forward pass -> x-sweep
    {
      // (the only sweep of this sub-step has no weight gradients: the forward stores a seventh of the record)
      XwOdeFwdJob jf = fwd_job(g->xT, g->start, g->u, g->Y, g->act, g->N, 1, (g->narrow >> 6) & 1);
      XW_TRY(xw_ode_fwd_multi(&jf, 1, g->t, s->theta, s->method, g->L, g->d, s->H, s->K, s->m, s->scal, stream));
      XwOdeBwdJob jx = bwd_job(g->xT, g->start, g->Y, g->act, g->N);
      jx.gx = g->gx; jx.gs = g->gs;
      XW_TRY(xw_ode_bwd_multi(&jx, 1, g->t, s->theta, s->method, g->L, g->d, s->H, s->K, s->m, 1 | adj | (((g->narrow >> 7) & 1) ? 16 : 0),
                              stream));
    }
    XW_HIP(hipEventRecord(sd->done, sd->s));
    stream = main_stream;
    XW_HIP(hipStreamWaitEvent((hipStream_t)main_stream, sd->done, 0));
    XW_TRY(contract(g, s, s->step_v, false, !shard, stream));
  } else {
    XW_ZERO(s->scal, 16);               // an empty share: zeros into both exchanges
  }
  if (shard) {
    // ---- exchange 1: I and sum v^2 must be global before the cotangent can be formed (+ the two factors of a pairwise group)
    XW_TRY(s->exchange(s->scal, 9, s->exchange_ctx, stream));
    if (g->pair_i) XW_TRY(xw_pair_fold(s->scal, g->Vol, g->Nglob, stream));
  }
  const int nsv = have_i ? xw_disc_bwd_slabs(g->N, g->L) : 0;
  if (have_i) {
    XW_TRY(xw_disc_cotangent(g->u, g->v, g->w, g->w_per_point, g->c, g->ckappa, g->f, g->h, g->N, g->L, g->Vol, g->Nglob, s->pollution,
                             g->s3_scale, s->scal, g->vbar, stream));
    if (g->tpp != nullptr) {
      XW_TRY(xw_disc_bwd(g->xvT_pts, nullptr, g->tpp, s->phi, g->vbar, g->N * g->L, 1, g->d, s->W, s->q, record, g->slab_v, stream));
    } else {
      XW_TRY(xw_disc_bwd(g->xvT, g->t, nullptr, s->phi, g->vbar, g->N, g->L, g->d, s->W, s->q, record, g->slab_v, stream));
    }
  }
  if (!shard) {
    XW_TRY(xw_adam(s->phi, g->slab_v, nsv, accum, nullptr, 0, nullptr, nullptr, s->m_v, s->v_v, s->step_v, -1, s->Pv, s->lr_v, s->beta1,
                   s->beta2, s->eps, s->grad_v, 0, 0, 0, nullptr, stream));
  } else {
    // ---- exchange 2: the packed gradient
    if (have_i) { XW_TRY(xw_slab_sum(g->slab_v, nsv, s->Pv, 0, s->grad_v, stream)); } else { XW_ZERO(s->grad_v, s->Pv); }
    XW_TRY(s->exchange(s->grad_v, s->Pv, s->exchange_ctx, stream));
    XW_TRY(xw_losses(s->scal, g->L, g->Lb > 0 ? g->Lb : 1, g->Vol, g->Nglob, g->Nbglob, s->alpha, 0.0, 0.0, s->step_v, stream));
    XW_TRY(xw_adam(s->phi, accum, accum != nullptr ? 1 : 0, s->grad_v, nullptr, 0, nullptr, nullptr, s->m_v, s->v_v, s->step_v, -1, s->Pv,
                   s->lr_v, s->beta1, s->beta2, s->eps, s->grad_v, 0, 0, 0, nullptr, stream));
  }
  if (accum != nullptr) {
    const hipError_t e = hipMemcpyAsync(accum, s->grad_v, sizeof(double) * s->Pv, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}
