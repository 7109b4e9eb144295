// xw_generic.hip -- the GENERIC-WIDTH path of both networks: any u_hidden_dim <= 64, u_hidden_hidden_dim <= 16, v_hidden_dim <= 128.
//
// The reference accepts any widths (src/model.py:30-43, 62-85, 130-138).  The MFMA kernels of xw_ode.hip / xw_disc.hip are built
// around the shipped shapes -- hidden-hidden activations are ONE 16-row tile, the test network's dVh maps four row tiles onto the
// four waves of a block -- and serve (20,10), (32,12) and W = 50, 64 (narrower networks run exactly inside them, zero-padded).
// Everything wider lands here: the same entry points (xw_ode_fwd_multi, xw_ode_bwd_multi, xw_disc_fwd, xw_disc_bwd: the dispatch
// is in xw_ode_abi.hip / xw_disc.hip), the same arguments, layouts and results, as plain per-path / per-point code on the vector
// ALU.  A SLOW path by construction -- one lane per path (stepper) or per point (test network), weights read through the scalar
// cache, no matrix instructions: two to three orders of magnitude below the MFMA kernels (profiles/r05_generic_widths.txt) --
// whose job is that a legal reference configuration trains instead of raising.
//
// Determinism: parameter gradients are reduced over the 16 paths of a slab (stepper) / the 64 points of a wave (test network) by
// a fixed butterfly of lane exchanges, entry by entry, and added to the slab 16 / 64 entries at a time by the lanes that hold them
// (an entry always by the same lane, in program order): no float atomics, same bits on every run and rank.
//
// Not here: the continuous adjoint (mode bit 3), the activation store (the sweeps recompute from the checkpoints Y), narrow tiles,
// priorities -- accepted and ignored where they are hints, XW_E_DIMS where they change the result.
#include "xw_common.h"
#include "xnwan.h"
#include "xw_generic.h"

namespace {
constexpr int GH = XWG_MAX_H, GK = XWG_MAX_K, GW = XWG_MAX_W, GM = XWG_MAX_M, GQ = XWG_MAX_Q;

__device__ __forceinline__ double gsum16(double x) {       // sum over the 16 lanes (paths) that share a slab
  x += __shfl_xor(x, 1);
  x += __shfl_xor(x, 2);
  x += __shfl_xor(x, 4);
  x += __shfl_xor(x, 8);
  return x;
}
__device__ __forceinline__ double gsum64(double x) {       // sum over the wave
  x = gsum16(x);
  x += __shfl_xor(x, 16);
  x += __shfl_xor(x, 32);
  return x;
}

// ---- u_theta -------------------------------------------------------------------------------------------------------------------
struct Net {
  const double* th;
  UOff o;
  int d, H, K, m;
};

// out[r] = init[r] + sum_c Wm[r ldw + c] x(c), r < rows -- four rows at a time (four independent chains, every x(c) used four times)
template <class FX>
__device__ __forceinline__ void matvec(const double* Wm, int ldw, int rows, int cols, FX x, const double* init, double* out) {
  for (int r = 0; r < rows; r += 4) {
    const int r1 = r + 1 < rows ? r + 1 : rows - 1, r2 = r + 2 < rows ? r + 2 : rows - 1, r3 = r + 3 < rows ? r + 3 : rows - 1;
    const double* w0 = Wm + (long)r * ldw;
    const double* w1 = Wm + (long)r1 * ldw;
    const double* w2 = Wm + (long)r2 * ldw;
    const double* w3 = Wm + (long)r3 * ldw;
    double a0 = init ? init[r] : 0.0, a1 = init ? init[r1] : 0.0, a2 = init ? init[r2] : 0.0, a3 = init ? init[r3] : 0.0;
#pragma unroll 4
    for (int c = 0; c < cols; ++c) {
      const double xc = x(c);
      a0 = fma(w0[c], xc, a0);
      a1 = fma(w1[c], xc, a1);
      a2 = fma(w2[c], xc, a2);
      a3 = fma(w3[c], xc, a3);
    }
    out[r] = a0;
    if (r + 1 < rows) out[r + 1] = a1;
    if (r + 2 < rows) out[r + 2] = a2;
    if (r + 3 < rows) out[r + 3] = a3;
  }
}
// out[c] = sum_r Wm[r ldw + c] x[r], c < cols (the transposed product) -- four adjacent columns at a time
__device__ __forceinline__ void matvecT(const double* Wm, int ldw, int rows, int cols, const double* x, double* out) {
  for (int c = 0; c < cols; c += 4) {
    const int c1 = c + 1 < cols ? c + 1 : cols - 1, c2 = c + 2 < cols ? c + 2 : cols - 1, c3 = c + 3 < cols ? c + 3 : cols - 1;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll 4
    for (int r = 0; r < rows; ++r) {
      const double xr = x[r];
      const double* w = Wm + (long)r * ldw;
      a0 = fma(w[c], xr, a0);
      a1 = fma(w[c1], xr, a1);
      a2 = fma(w[c2], xr, a2);
      a3 = fma(w[c3], xr, a3);
    }
    out[c] = a0;
    if (c + 1 < cols) out[c + 1] = a1;
    if (c + 2 < cols) out[c + 2] = a2;
    if (c + 3 < cols) out[c + 3] = a3;
  }
}

// F([x, t, y]) (src/model.py:153-156, 130-141); xproj[k] = Win[k, 0..d) x + Win.b[k] is hoisted (x does not move along a path)
// zs (or null): pre-activations of every layer, [m][GK], for the vector-Jacobian product
__device__ void field_eval(const Net& n, const double* xproj, double t, const double* y, double* out, double* zs) {
  const double* Win = n.th + n.o.Win;
  const int ld = n.o.ldin, K = n.K;
  double z[GK], z2[GK];
  for (int k = 0; k < K; ++k) z2[k] = fma(Win[k * ld + n.d], t, xproj[k]);
  matvec(Win + n.d + 1, ld, K, n.H, [&](int j) { return y[j]; }, z2, z);
  if (zs)
    for (int k = 0; k < K; ++k) zs[k] = z[k];
  for (int l = 1; l < n.m; ++l) {
    matvec(n.th + n.o.Wh, K, K, K, [&](int kk) { return z[kk] > 0.0 ? z[kk] : 0.0; }, n.th + n.o.Whb, z2);
    for (int k = 0; k < K; ++k) {
      z[k] = z2[k];
      if (zs) zs[l * GK + k] = z2[k];
    }
  }
  for (int k = 0; k < K; ++k) z[k] = xw_tanh(z[k]);
  matvec(n.th + n.o.Wo, K, n.H, K, [&](int k) { return z[k]; }, n.th + n.o.Wob, out);
}

// slab[e0 + i] += sum over the 16 paths of the group of term(i), i < n.  Sixteen entries at a time: every lane of the group ends
// up holding the total of ONE entry (a fixed butterfly of lane exchanges per entry) and the group adds them with one coalesced
// read-modify-write -- an entry is always touched by the same lane, in program order: deterministic, no atomics, and the memory
// round trip is paid once per 16 entries (per entry it was 0.5 us: 120 of the 150 ms of a sweep at (64, 16)).
template <class F>
__device__ __forceinline__ void gadd_run(double* slab, int e0, int n, bool active, F term) {
  const int l16 = threadIdx.x & 15;
  for (int c = 0; c < n; c += 16) {
    double mine = 0.0;
    for (int i = 0; i < 16; ++i) {
      if (c + i >= n) break;
      const double s_ = gsum16(active ? term(c + i) : 0.0);
      if (l16 == i) mine = s_;
    }
    if (c + l16 < n) slab[e0 + c + l16] += mine;
  }
}

// a^T dF/d(y, theta) at (t, yin): gy[H] (overwritten), Sx[K] += cotangent of the input layer's pre-activation (the x columns and
// the bias of Win are contracted once per sweep from it), parameter gradients into the group's slab (or none: slab == null)
__device__ void field_vjp(const Net& n, const double* xproj, double t, const double* yin, const double* a, double* gy, double* Sx,
                          double* slab, bool active) {
  double zs[GM * GK], out[GH];
  field_eval(n, xproj, t, yin, out, zs);
  const int K = n.K, H = n.H, ld = n.o.ldin;
  const double* Wo = n.th + n.o.Wo;
  const double* Wh = n.th + n.o.Wh;
  const double* Win = n.th + n.o.Win;
  double dz[GK], dzp[GK], th[GK];
  for (int k = 0; k < K; ++k) th[k] = xw_tanh(zs[(n.m - 1) * GK + k]);
  matvecT(Wo, K, H, K, a, dz);
  for (int k = 0; k < K; ++k) dz[k] *= 1.0 - th[k] * th[k];
  if (slab) {
    for (int h = 0; h < H; ++h) {
      const double ah = a[h];
      gadd_run(slab, n.o.Wo + h * K, K, active, [&](int k) { return ah * th[k]; });
    }
    gadd_run(slab, n.o.Wob, H, active, [&](int h) { return a[h]; });
  }
  for (int l = n.m - 1; l >= 1; --l) {
    const double* zp = zs + (l - 1) * GK;
    if (slab) {
      for (int k = 0; k < K; ++k) {
        const double dk = dz[k];
        gadd_run(slab, n.o.Wh + k * K, K, active, [&](int kk) { return dk * (zp[kk] > 0.0 ? zp[kk] : 0.0); });
      }
      gadd_run(slab, n.o.Whb, K, active, [&](int k) { return dz[k]; });
    }
    matvecT(Wh, K, K, K, dz, dzp);
    for (int k = 0; k < K; ++k) dz[k] = zp[k] > 0.0 ? dzp[k] : 0.0;
  }
  if (slab) {
    for (int k = 0; k < K; ++k) {
      const double dk = dz[k];
      // columns d (the time) and d + 1 .. d + H (the state) of row k are contiguous
      gadd_run(slab, n.o.Win + k * ld + n.d, 1 + H, active, [&](int c) { return dk * (c == 0 ? t : yin[c - 1]); });
    }
  }
  for (int k = 0; k < K; ++k) Sx[k] += dz[k];
  matvecT(Win + n.d + 1, ld, K, H, dz, gy);
}

__device__ void lift(const Net& n, double s, double* pre0, double* pre2, double* y) {   // y0 = IL(start), src/model.py:78,97
  const double* th = n.th;
  const int H = n.H;
  for (int i = 0; i < H; ++i) pre0[i] = fma(th[n.o.IL0w + i], s, th[n.o.IL0b + i]);
  matvec(th + n.o.IL2w, H, H, H, [&](int j) { return pre0[j] > 0.0 ? pre0[j] : 0.0; }, th + n.o.IL2b, pre2);
  matvec(th + n.o.IL4w, H, H, H, [&](int j) { return pre2[j] > 0.0 ? pre2[j] : 0.0; }, th + n.o.IL4b, y);
}

__device__ void x_projection(const Net& n, const double* xT, int N, int path, double* xproj) {
  const double* Win = n.th + n.o.Win;
  for (int k = 0; k < n.K; ++k) {
    double acc = n.th[n.o.Winb + k];
    for (int i = 0; i < n.d; ++i) acc = fma(Win[k * n.o.ldin + i], xT[(long)i * N + path], acc);
    xproj[k] = acc;
  }
}

// one step of the fixed-grid schemes (oracle.odeint_fixed: euler, midpoint, the 3/8-rule rk4) from y at t0 over dt
__device__ void rk_step(const Net& n, int method, const double* xproj, double t0, double dt, double* y) {
  const int H = n.H;
  double k1[GH], k2[GH], k3[GH], tmp[GH];
  field_eval(n, xproj, t0, y, k1, nullptr);
  if (method == 0) {
    for (int j = 0; j < H; ++j) y[j] = fma(dt, k1[j], y[j]);
  } else if (method == 1) {
    for (int j = 0; j < H; ++j) tmp[j] = fma(k1[j], dt / 2, y[j]);
    field_eval(n, xproj, t0 + dt / 2, tmp, k2, nullptr);
    for (int j = 0; j < H; ++j) y[j] = fma(dt, k2[j], y[j]);
  } else {
    for (int j = 0; j < H; ++j) tmp[j] = y[j] + dt * k1[j] / 3;
    field_eval(n, xproj, t0 + dt / 3, tmp, k2, nullptr);
    for (int j = 0; j < H; ++j) tmp[j] = y[j] + dt * (k2[j] - k1[j] / 3);
    field_eval(n, xproj, t0 + 2 * dt / 3, tmp, k3, nullptr);
    for (int j = 0; j < H; ++j) tmp[j] = y[j] + dt * (k1[j] - k2[j] + k3[j]);
    double k4[GH];
    field_eval(n, xproj, t0 + dt, tmp, k4, nullptr);
    for (int j = 0; j < H; ++j) y[j] = y[j] + dt * (k1[j] + 3 * (k2[j] + k3[j]) + k4[j]) / 8;
  }
}

__global__ void __launch_bounds__(64) kg_ode_fwd(XwOdeFwdJob job, const double* __restrict__ tf, const double* __restrict__ theta,
                                                  int method, int L, int d, int H, int K, int m) {
  const int path = blockIdx.x * 64 + threadIdx.x;
  if (path >= job.N) return;
  const int N = job.N;
  Net n = {theta, u_offsets(d, H, K), d, H, K, m};
  double xproj[GK], y[GH], p0[GH], p2[GH];
  x_projection(n, job.xT, N, path, xproj);
  lift(n, job.start[path], p0, p2, y);
  const double* flw = theta + n.o.FLw;
  for (int l = 0; l < L; ++l) {
    if (l > 0) rk_step(n, method, xproj, tf[l - 1], tf[l] - tf[l - 1], y);
    double acc = theta[n.o.FLb];
    for (int j = 0; j < H; ++j) acc = fma(flw[j], y[j], acc);
    job.u[(long)l * N + path] = acc;
    if (job.Y)
      for (int j = 0; j < H; ++j) job.Y[((long)l * H + j) * N + path] = y[j];
  }
}

// cotangent on u at (l, path): stored, all ones, or one of the residual forms of XwOdeBwdJob
__device__ double cot_u(const XwOdeBwdJob& j, int l, int L, int path) {
  const long p = (long)l * j.N + path;
  if (j.res_u == nullptr) return j.ubar ? j.ubar[p] : 1.0;
  if (j.res_first_only == 2) {
    const double u = j.res_u[p], v = j.res_ref[p];
    const double w = j.res_w_per_point ? j.res_w[p] : j.res_w[path];
    const double dcu = j.res_c != nullptr ? j.res_c[p] + u * j.res_cp[p] : j.res_kappa2 * u;
    double g = j.res_coef * dcu * v * w;
    if (l == L - 1) g += j.res_base * v;
    return g;
  }
  if (j.res_first_only == 1) return l == 0 ? j.res_base + j.res_coef * (j.res_u[p] - j.res_ref[path]) : j.res_base;
  return j.res_base + j.res_coef * (j.res_u[p] - j.res_ref[p]);
}

// reverse sweep through the discrete stepper (include/xnwan.h: xw_ode_bwd, mode bits 0..2), recomputing from the checkpoints Y
__global__ void __launch_bounds__(64) kg_ode_bwd(XwOdeBwdJob job, const double* __restrict__ tf, const double* __restrict__ theta,
                                                  int method, int L, int d, int H, int K, int m, int mode) {
  const int N = job.N;
  const int raw = blockIdx.x * 64 + threadIdx.x;
  const bool active = raw < N;
  const int path = active ? raw : N - 1;                       // (lanes past the end walk along with the last path, adding zeros)
  const bool want_x = (mode & 1) != 0, ones_x = (mode & 4) != 0;
  double* slab = (mode & 2) ? job.gslab + (long)(raw >> 4) * u_offsets(d, H, K).total : nullptr;
  if (slab != nullptr && (raw >> 4) * 16 >= N) slab = nullptr;  // (a group entirely past the end owns no slab)
  Net n = {theta, u_offsets(d, H, K), d, H, K, m};
  double xproj[GK], Sx[GK], lam[GH], y[GH], gy[GH], a[GH];
  x_projection(n, job.xT, N, path, xproj);
  for (int k = 0; k < K; ++k) Sx[k] = 0.0;
  for (int j = 0; j < H; ++j) lam[j] = 0.0;
  const double* flw = theta + n.o.FLw;
  for (int l = L - 1; l >= 1; --l) {
    const double ub = cot_u(job, l, L, path);
    for (int j = 0; j < H; ++j) {
      y[j] = job.Y[((long)l * H + j) * N + path];
      lam[j] = fma(flw[j], ub, lam[j]);
    }
    if (slab) {
      gadd_run(slab, n.o.FLw, H, active, [&](int j) { return ub * y[j]; });
      gadd_run(slab, n.o.FLb, 1, active, [&](int) { return ub; });
    }
    // y_l = step(y_{l-1}): lam becomes the cotangent of y_{l-1}
    const double t0 = tf[l - 1], dt = tf[l] - tf[l - 1];
    for (int j = 0; j < H; ++j) y[j] = job.Y[((long)(l - 1) * H + j) * N + path];
    if (method == 0) {
      for (int j = 0; j < H; ++j) a[j] = dt * lam[j];
      field_vjp(n, xproj, t0, y, a, gy, Sx, slab, active);
      for (int j = 0; j < H; ++j) lam[j] += gy[j];
    } else if (method == 1) {
      double k1[GH], ym[GH];
      field_eval(n, xproj, t0, y, k1, nullptr);
      for (int j = 0; j < H; ++j) {
        ym[j] = fma(k1[j], dt / 2, y[j]);
        a[j] = dt * lam[j];
      }
      field_vjp(n, xproj, t0 + dt / 2, ym, a, gy, Sx, slab, active);
      for (int j = 0; j < H; ++j) {
        lam[j] += gy[j];
        a[j] = (dt / 2) * gy[j];
      }
      field_vjp(n, xproj, t0, y, a, gy, Sx, slab, active);
      for (int j = 0; j < H; ++j) lam[j] += gy[j];
    } else {
      double k1[GH], k2[GH], k3[GH], Y2[GH], Y3[GH], Y4[GH], g4[GH], g3[GH], g2[GH];
      field_eval(n, xproj, t0, y, k1, nullptr);
      for (int j = 0; j < H; ++j) Y2[j] = y[j] + dt * k1[j] / 3;
      field_eval(n, xproj, t0 + dt / 3, Y2, k2, nullptr);
      for (int j = 0; j < H; ++j) Y3[j] = y[j] + dt * (k2[j] - k1[j] / 3);
      field_eval(n, xproj, t0 + 2 * dt / 3, Y3, k3, nullptr);
      for (int j = 0; j < H; ++j) Y4[j] = y[j] + dt * (k1[j] - k2[j] + k3[j]);
      for (int j = 0; j < H; ++j) a[j] = (dt / 8) * lam[j];
      field_vjp(n, xproj, t0 + dt, Y4, a, g4, Sx, slab, active);
      for (int j = 0; j < H; ++j) a[j] = (3 * dt / 8) * lam[j] + dt * g4[j];
      field_vjp(n, xproj, t0 + 2 * dt / 3, Y3, a, g3, Sx, slab, active);
      for (int j = 0; j < H; ++j) a[j] = (3 * dt / 8) * lam[j] - dt * g4[j] + dt * g3[j];
      field_vjp(n, xproj, t0 + dt / 3, Y2, a, g2, Sx, slab, active);
      for (int j = 0; j < H; ++j) a[j] = (dt / 8) * lam[j] + dt * g4[j] - (dt / 3) * g3[j] + (dt / 3) * g2[j];
      field_vjp(n, xproj, t0, y, a, gy, Sx, slab, active);
      for (int j = 0; j < H; ++j) lam[j] += g4[j] + g3[j] + g2[j] + gy[j];
    }
  }
  // l = 0: read-out, then the lift 1 -> H -> H -> H (src/model.py:78); with mode bit 2 the x-side outputs are those of the
  // ALL-ONES cotangent (the helper backward of src/loss.py:55) while the parameter gradients use the job's own
  const double ub0 = cot_u(job, 0, L, path);
  double p0[GH], p2[GH];
  lift(n, job.start[path], p0, p2, y);
  if (slab) {
    gadd_run(slab, n.o.FLw, H, active, [&](int j) { return ub0 * y[j]; });
    gadd_run(slab, n.o.FLb, 1, active, [&](int) { return ub0; });
  }
  const double s = job.start[path];
  for (int pass = 0; pass < 2; ++pass) {
    // pass 0: parameter gradients (cotangent ub0); pass 1: d/d start (cotangent 1 with mode bit 2, else ub0)
    if (pass == 0 && !slab) continue;
    if (pass == 1 && !(want_x && job.gs != nullptr)) continue;
    const double ub = pass == 1 && ones_x ? 1.0 : ub0;
    double l0[GH], dh2[GH], dh1[GH];
    for (int j = 0; j < H; ++j) l0[j] = fma(flw[j], ub, lam[j]);
    matvecT(theta + n.o.IL4w, H, H, H, l0, dh2);
    for (int j = 0; j < H; ++j) dh2[j] = p2[j] > 0.0 ? dh2[j] : 0.0;
    matvecT(theta + n.o.IL2w, H, H, H, dh2, dh1);
    for (int j = 0; j < H; ++j) dh1[j] = p0[j] > 0.0 ? dh1[j] : 0.0;
    if (pass == 0) {
      for (int i = 0; i < H; ++i) {
        const double li = l0[i], di = dh2[i];
        gadd_run(slab, n.o.IL4w + i * H, H, active, [&](int j) { return li * (p2[j] > 0.0 ? p2[j] : 0.0); });
        gadd_run(slab, n.o.IL2w + i * H, H, active, [&](int j) { return di * (p0[j] > 0.0 ? p0[j] : 0.0); });
      }
      gadd_run(slab, n.o.IL4b, H, active, [&](int i) { return l0[i]; });
      gadd_run(slab, n.o.IL2b, H, active, [&](int i) { return dh2[i]; });
      gadd_run(slab, n.o.IL0w, H, active, [&](int i) { return dh1[i] * s; });
      gadd_run(slab, n.o.IL0b, H, active, [&](int i) { return dh1[i]; });
    } else if (active) {
      double acc = 0.0;
      for (int i = 0; i < H; ++i) acc = fma(theta[n.o.IL0w + i], dh1[i], acc);
      job.gs[path] = acc;
    }
  }
  // the x columns and the bias of the input layer, from the summed cotangent of its pre-activation
  const double* Win = theta + n.o.Win;
  if (slab) {
    gadd_run(slab, n.o.Winb, K, active, [&](int k) { return Sx[k]; });
    for (int k = 0; k < K; ++k) {
      const double sk = Sx[k];
      gadd_run(slab, n.o.Win + k * n.o.ldin, d, active, [&](int i) { return sk * job.xT[(long)i * N + path]; });
    }
  }
  if (want_x && job.gx != nullptr && active)
    for (int i = 0; i < d; ++i) {
      double acc = 0.0;
      for (int k = 0; k < K; ++k) acc = fma(Win[k * n.o.ldin + i], Sx[k], acc);
      job.gx[(long)i * N + path] = acc;
    }
}

// ---- v_phi ---------------------------------------------------------------------------------------------------------------------
// point -> (time, path): path mode p = l N + n, point mode (tpp) p = n
__device__ __forceinline__ void locate_pt(long p, int N, const double* tf, const double* tpp, double& t, int& nidx) {
  if (tpp != nullptr) {
    nidx = (int)p;
    t = tpp[p];
  } else {
    const int l = (int)(p / N);
    nidx = (int)(p - (long)l * N);
    t = tf[l];
  }
}

__global__ void __launch_bounds__(64) kg_disc_fwd(const double* __restrict__ xT, const double* __restrict__ tf,
                                                   const double* __restrict__ tpp, const double* __restrict__ ph, int N, int L,
                                                   int d, int W, int q, double* __restrict__ v, double* __restrict__ vt,
                                                   double* __restrict__ gxv, double* __restrict__ gtv, int ngrad,
                                                   double* __restrict__ act, long cols) {
  const long P = (long)N * L;
  const long p = (long)blockIdx.x * 64 + threadIdx.x;
  if (p >= P) return;
  const VOff o = v_offsets(d, W);
  double t;
  int nidx;
  locate_pt(p, N, tf, tpp, t, nidx);
  double a[GW], ad[GW], nw[GW], nd[GW];
  unsigned long long mk[GQ][(GW + 63) / 64];
  const bool grad = gxv != nullptr && p < ngrad;
  for (int k = 0; k < W; ++k) {
    double acc = fma(ph[o.Vin + k * o.ldin], t, ph[o.Vinb + k]);
    for (int i = 0; i < d; ++i) acc = fma(ph[o.Vin + k * o.ldin + 1 + i], xT[(long)i * N + nidx], acc);
    a[k] = acc;
    ad[k] = ph[o.Vin + k * o.ldin];
  }
  for (int j = 0; j < q; ++j) {
    if (grad)
      for (int w_ = 0; w_ < (W + 63) / 64; ++w_) mk[j][w_] = 0ull;
    for (int k = 0; k < W; ++k) {                  // relu and the tangent's gate: relu'(0) = 0 (torch)
      const bool open = a[k] > 0.0;
      a[k] = open ? a[k] : 0.0;
      ad[k] = open ? ad[k] : 0.0;
      if (grad && open) mk[j][k >> 6] |= 1ull << (k & 63);
      if (act) act[((long)j * W + k) * cols + p] = a[k];
    }
    for (int k = 0; k < W; k += 4) {                // four output rows at a time: a loaded activation feeds 8 multiply-adds
      const int k1 = k + 1 < W ? k + 1 : W - 1, k2 = k + 2 < W ? k + 2 : W - 1, k3 = k + 3 < W ? k + 3 : W - 1;
      const double* w0 = ph + o.Vh + k * W;
      const double* w1 = ph + o.Vh + k1 * W;
      const double* w2 = ph + o.Vh + k2 * W;
      const double* w3 = ph + o.Vh + k3 * W;
      double s0 = ph[o.Vhb + k], s1 = ph[o.Vhb + k1], s2 = ph[o.Vhb + k2], s3 = ph[o.Vhb + k3];
      double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
#pragma unroll 4
      for (int kk = 0; kk < W; ++kk) {
        const double av = a[kk], adv = ad[kk];
        s0 = fma(w0[kk], av, s0); t0 = fma(w0[kk], adv, t0);
        s1 = fma(w1[kk], av, s1); t1 = fma(w1[kk], adv, t1);
        s2 = fma(w2[kk], av, s2); t2 = fma(w2[kk], adv, t2);
        s3 = fma(w3[kk], av, s3); t3 = fma(w3[kk], adv, t3);
      }
      nw[k] = s0; nd[k] = t0;
      if (k + 1 < W) { nw[k + 1] = s1; nd[k + 1] = t1; }
      if (k + 2 < W) { nw[k + 2] = s2; nd[k + 2] = t2; }
      if (k + 3 < W) { nw[k + 3] = s3; nd[k + 3] = t3; }
    }
    for (int k = 0; k < W; ++k) {
      a[k] = nw[k];
      ad[k] = nd[k];
    }
  }
  double sv = ph[o.Vob], sd = 0.0;
  for (int k = 0; k < W; ++k) {
    const double th = xw_tanh(a[k]), vo = ph[o.Vo + k];
    if (act) act[((long)q * W + k) * cols + p] = th;
    sv = fma(vo, th, sv);
    sd = fma(vo * (1.0 - th * th), ad[k], sd);
    a[k] = vo * (1.0 - th * th);                   // cotangent of a_q for d(sum v)/d(input)
  }
  v[p] = sv;
  if (vt) vt[p] = sd;
  if (grad) {
    for (int j = q - 1; j >= 0; --j) {
      for (int kk = 0; kk < W; ++kk) {
        double acc = 0.0;
        for (int k = 0; k < W; ++k) acc = fma(ph[o.Vh + k * W + kk], a[k], acc);
        nw[kk] = ((mk[j][kk >> 6] >> (kk & 63)) & 1ull) ? acc : 0.0;
      }
      for (int k = 0; k < W; ++k) a[k] = nw[k];
    }
    for (int i = 0; i < d; ++i) {
      double acc = 0.0;
      for (int k = 0; k < W; ++k) acc = fma(ph[o.Vin + k * o.ldin + 1 + i], a[k], acc);
      gxv[(long)i * ngrad + p] = acc;
    }
    if (gtv) {
      double acc = 0.0;
      for (int k = 0; k < W; ++k) acc = fma(ph[o.Vin + k * o.ldin], a[k], acc);
      gtv[p] = acc;
    }
  }
}

// slab[e0 + i] += sum over the 64 points of the wave of term(i), i < n: as gadd_run, 64 entries per coalesced read-modify-write
template <class F>
__device__ __forceinline__ void wadd_run(double* slab, int e0, int n, F term) {
  const int lane = threadIdx.x;
  for (int c = 0; c < n; c += 64) {
    double mine = 0.0;
    for (int i = 0; i < 64; ++i) {
      if (c + i >= n) break;
      const double s_ = gsum64(term(c + i));
      if (lane == i) mine = s_;
    }
    if (c + lane < n) slab[e0 + c + lane] += mine;
  }
}

// parameter gradient of <vbar, v> from the record of kg_disc_fwd; one wave per block, one slab per block, 64 points per pass.
// The reverse chain runs first and keeps the cotangent of every layer (per lane, scratch); dVh is then formed entry by entry over
// ALL layers at once -- sum_j delta_{j+1}[k] r_j[kk] per lane, one butterfly over the 64 points, one coalesced slab update per
// 64 entries, four rows of dVh per pass over the record -- instead of a memory round trip per entry and layer (387 -> 82 ms at W = 128,
// headline sample).
__global__ void __launch_bounds__(64) kg_disc_bwd(const double* __restrict__ xT, const double* __restrict__ tf,
                                                   const double* __restrict__ tpp, const double* __restrict__ ph,
                                                   const double* __restrict__ vbar, int N, int L, int d, int W, int q,
                                                   const double* __restrict__ act, long cols, double* __restrict__ gslab) {
  const long P = (long)N * L;
  const long nsuper = (P + 63) / 64;
  const VOff o = v_offsets(d, W);
  double* slab = gslab + (long)blockIdx.x * o.total;
  for (long st = blockIdx.x; st < nsuper; st += gridDim.x) {
    const long raw = st * 64 + threadIdx.x;
    const bool valid = raw < P;
    const long p = valid ? raw : P - 1;
    double t;
    int nidx;
    locate_pt(p, N, tf, tpp, t, nidx);
    const double vb = valid ? (vbar ? vbar[p] : 1.0) : 0.0;
    double dls[(GQ + 1) * GW];                                   // dls[j * GW + k]: cotangent of a_j[k] (j = q: of the last pre-activation)
    double* dq = dls + q * GW;
    for (int k = 0; k < W; ++k) {
      const double th = act[((long)q * W + k) * cols + p];
      dq[k] = ph[o.Vo + k] * (1.0 - th * th) * vb;
    }
    wadd_run(slab, o.Vo, W, [&](int k) { return vb * act[((long)q * W + k) * cols + p]; });
    wadd_run(slab, o.Vob, 1, [&](int) { return vb; });
    for (int j = q - 1; j >= 0; --j) {
      const double* r = act + (long)j * W * cols + p;            // r[kk * cols]: input kk of tied layer j at this point
      const double* dn = dls + (j + 1) * GW;
      double* dj = dls + j * GW;
      for (int kk = 0; kk < W; kk += 4) {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        const int c1 = kk + 1 < W ? kk + 1 : W - 1, c2 = kk + 2 < W ? kk + 2 : W - 1, c3 = kk + 3 < W ? kk + 3 : W - 1;
#pragma unroll 4
        for (int k = 0; k < W; ++k) {
          const double dk = dn[k];
          const double* row = ph + o.Vh + k * W;
          a0 = fma(row[kk], dk, a0);
          a1 = fma(row[c1], dk, a1);
          a2 = fma(row[c2], dk, a2);
          a3 = fma(row[c3], dk, a3);
        }
        dj[kk] = r[(long)kk * cols] > 0.0 ? a0 : 0.0;
        if (kk + 1 < W) dj[kk + 1] = r[(long)(kk + 1) * cols] > 0.0 ? a1 : 0.0;
        if (kk + 2 < W) dj[kk + 2] = r[(long)(kk + 2) * cols] > 0.0 ? a2 : 0.0;
        if (kk + 3 < W) dj[kk + 3] = r[(long)(kk + 3) * cols] > 0.0 ? a3 : 0.0;
      }
    }
    // dVh[k][kk] += sum_j delta_{j+1}[k] r_j[kk];  dVh.b[k] += sum_j delta_{j+1}[k].  FOUR rows k at a time: the q record values
    // r_j[kk] of an input kk are loaded once per four rows (the record of a 64-point pass, 0.6 MB at W = 128, does not stay in
    // cache across the passes of all the waves of an XCD: row by row it was re-read W times from HBM)
    for (int k = 0; k < W; k += 4) {
      double dk[4][GQ];
#pragma unroll
      for (int j = 0; j < GQ; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) dk[i][j] = (j < q && k + i < W) ? dls[(j + 1) * GW + k + i] : 0.0;
      const int lane = threadIdx.x;
      for (int c = 0; c < W; c += 64) {
        double mine[4] = {0.0, 0.0, 0.0, 0.0};
        for (int i = 0; i < 64; ++i) {
          if (c + i >= W) break;
          double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
          for (int j = 0; j < GQ; ++j) {
            if (j < q) {
              const double rj = act[((long)j * W + c + i) * cols + p];
              a0 = fma(dk[0][j], rj, a0);
              a1 = fma(dk[1][j], rj, a1);
              a2 = fma(dk[2][j], rj, a2);
              a3 = fma(dk[3][j], rj, a3);
            }
          }
          a0 = gsum64(a0); a1 = gsum64(a1); a2 = gsum64(a2); a3 = gsum64(a3);
          if (lane == i) { mine[0] = a0; mine[1] = a1; mine[2] = a2; mine[3] = a3; }
        }
        if (c + lane < W) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (k + i < W) slab[o.Vh + (k + i) * W + c + lane] += mine[i];
        }
      }
    }
    wadd_run(slab, o.Vhb, W, [&](int k) {
      double acc = 0.0;
      for (int j = 0; j < q; ++j) acc += dls[(j + 1) * GW + k];
      return acc;
    });
    // input layer: row k of dVin = delta_0[k] [t, x], contiguous; dVin.b = delta_0
    wadd_run(slab, o.Vinb, W, [&](int k) { return dls[k]; });
    for (int k = 0; k < W; ++k) {
      const double d0 = dls[k];
      wadd_run(slab, o.Vin + k * o.ldin, 1 + d, [&](int c) { return d0 * (c == 0 ? t : xT[(long)(c - 1) * N + nidx]); });
    }
  }
}

}  // namespace

// ---- entry points behind the public ABI (hidden: the library exports xw_* only) -------------------------------------------------
int xwg_ode_ok(int d, int H, int K, int m) { return H >= 1 && H <= GH && K >= 1 && K <= GK && m >= 1 && m <= GM && d >= 1 && d + 2 <= 128; }
int xwg_disc_ok(int d, int W, int q) { return W >= 1 && W <= GW && q >= 0 && q <= GQ && d >= 1 && d + 2 <= 128; }

int xwg_ode_fwd_multi(const XwOdeFwdJob* jobs, int njobs, const double* t, const double* theta, int method, int L, int d, int H,
                      int K, int m, double* zero16, void* stream) {
  if (!jobs || njobs < 1 || !t || !theta || L < 1 || method < 0 || method > 2) return XW_E_ARG;
  if (!xwg_ode_ok(d, H, K, m)) return XW_E_DIMS;
  hipStream_t s = (hipStream_t)stream;
  if (zero16 != nullptr) {
    const hipError_t e = hipMemsetAsync(zero16, 0, 16 * sizeof(double), s);
    if (e != hipSuccess) return (int)e;
  }
  for (int i = 0; i < njobs; ++i) {
    const XwOdeFwdJob& j = jobs[i];
    if (!j.xT || !j.start || !j.u || j.N < 1) return XW_E_ARG;
    hipLaunchKernelGGL(kg_ode_fwd, dim3((j.N + 63) / 64), dim3(64), 0, s, j, t, theta, method, L, d, H, K, m);
  }
  return xw_launch_status();
}

int xwg_ode_bwd_multi(const XwOdeBwdJob* jobs, int njobs, const double* t, const double* theta, int method, int L, int d, int H,
                      int K, int m, int mode, void* stream) {
  if (!jobs || njobs < 1 || !t || !theta || L < 1 || method < 0 || method > 2 || (mode & 3) == 0) return XW_E_ARG;
  if (!xwg_ode_ok(d, H, K, m) || (mode & 8)) return XW_E_DIMS;          // (no continuous adjoint at the generic widths)
  if ((mode & 4) && (mode & 3) != 3) return XW_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  const long P = u_offsets(d, H, K).total;
  for (int i = 0; i < njobs; ++i) {
    const XwOdeBwdJob& j = jobs[i];
    if (!j.xT || !j.start || !j.Y || j.N < 1) return XW_E_ARG;
    if (j.res_u != nullptr && j.ubar != nullptr) return XW_E_ARG;
    if ((mode & 2) && !j.gslab) return XW_E_ARG;
    if ((mode & 1) && !(mode & 4) && (!j.gx || !j.gs)) return XW_E_ARG;
    if (mode & 2) {
      const hipError_t e = hipMemsetAsync(j.gslab, 0, sizeof(double) * P * ((j.N + 15) / 16), s);
      if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kg_ode_bwd, dim3((j.N + 63) / 64), dim3(64), 0, s, j, t, theta, method, L, d, H, K, m, mode);
  }
  return xw_launch_status();
}

int xwg_disc_fwd(const double* xT, const double* t, const double* tpp, const double* phi, int N, int L, int d, int W, int q,
                 double* v, double* vt, double* gxv, double* gtv, int ngrad, double* act, void* stream) {
  if (!xwg_disc_ok(d, W, q)) return XW_E_DIMS;
  const long P = (long)N * L;
  const long cols = (P + 15) / 16 * 16;
  hipLaunchKernelGGL(kg_disc_fwd, dim3((unsigned)((P + 63) / 64)), dim3(64), 0, (hipStream_t)stream, xT, t, tpp, phi, N, L, d, W, q, v,
                     vt, gxv, gtv, ngrad, act, cols);
  return xw_launch_status();
}

int xwg_disc_bwd(const double* xT, const double* t, const double* tpp, const double* phi, const double* vbar, int N, int L, int d,
                 int W, int q, const double* act, double* gslab, int nslab, void* stream) {
  if (!xwg_disc_ok(d, W, q)) return XW_E_DIMS;
  if (!act) return XW_E_DIMS;                                            // (from the record only; kernels.disc_bwd stores one first)
  const long P = (long)N * L;
  const long cols = (P + 15) / 16 * 16;
  const hipError_t e = hipMemsetAsync(gslab, 0, sizeof(double) * (size_t)v_offsets(d, W).total * nslab, (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kg_disc_bwd, dim3(nslab), dim3(64), 0, (hipStream_t)stream, xT, t, tpp, phi, vbar, N, L, d, W, q, act, cols, gslab);
  return xw_launch_status();
}
