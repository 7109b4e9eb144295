// xw_ode_abi.hip -- public entry points of the stepper (include/xnwan.h): argument checks that do not depend on the
// width, and the choice of the object compiled for (H, K) (xw_ode.hip, one object per width).
#include "xw_common.h"
#include "xnwan.h"
#include "xw_generic.h"

#define XW_ODE_WIDTHS(X) X(20, 10) X(32, 12)      /* keep in step with the Makefile and kernels.ODE_WIDTHS */
/* ... and the WIDE container (64, 16) (round 6; xw_ode.hip -DXW_ODE_WIDE16: the field on v_mfma_f64_16x16x4, one wave per tile --
 * the sweep with weight gradients from the store: chain wave + partner wave --, no narrow tiles, depths 1..10): same entry points */
#define XW_WIDE_H 64
#define XW_WIDE_K 16
extern "C" int xw_ode_fwd_multi_w64_16(const XwOdeFwdJob*, int, const double*, const double*, int, int, int, int, double*, void*);
extern "C" int xw_ode_bwd_multi_w64_16(const XwOdeBwdJob*, int, const double*, const double*, int, int, int, int, int, void*);

#define DECL(H, K)                                                                                                         \
  extern "C" int xw_ode_fwd_multi_w##H##_##K(const XwOdeFwdJob*, int, const double*, const double*, int, int, int, int, double*, void*); \
  extern "C" int xw_ode_bwd_multi_w##H##_##K(const XwOdeBwdJob*, int, const double*, const double*, int, int, int, int, int, void*);
XW_ODE_WIDTHS(DECL)
#undef DECL

extern "C" int xw_ode_bwd_slabs(int N) { return (N + 15) / 16; }

static bool width_compiled(int H, int K) {
#define TEST(HH, KK) if (H == HH && K == KK) return true;
  XW_ODE_WIDTHS(TEST)
#undef TEST
  return false;
}

extern "C" int xw_ode_act_rows(int method, int H, int K, int m) {
  if (H == XW_WIDE_H && K == XW_WIDE_K) {                             // the wide container
    const int S_ = method == 0 ? 1 : method == 1 ? 2 : 0;
    if (!xwg_ode_ok(1, H, K, m)) return XW_E_DIMS;
    // (layer inputs + the stage inputs + the ReLU-mask words of every stage: 4 (m - 1) bits per lane, two words at depth 10)
    return (S_ == 0 || m > XW_ODE_MAX_LAYERS) ? 0 : S_ * m * K + (S_ - 1) * H + (4 * (m - 1) > 32 ? 4 : 2) * S_;
  }
  if ((!width_compiled(H, K) || m > XW_ODE_MAX_LAYERS) && xwg_ode_ok(1, H, K, m)) return 0;   // generic widths / depths (xw_generic.hip): the sweeps recompute
  if (!width_compiled(H, K) || m < 1 || m > XW_ODE_MAX_LAYERS) return XW_E_DIMS;
  const int S = method == 0 ? 1 : method == 1 ? 2 : 0;           // rk4: the sweeps recompute
  return S == 0 ? 0 : S * m * K + (S - 1) * H + 2 * S;             // (+ the ReLU-mask words of every stage)
}

extern "C" int xw_ode_fwd_multi(const XwOdeFwdJob* jobs, int njobs, const double* t, const double* theta, int method, int L,
                                int d, int H, int K, int m, double* zero16, void* stream) {
#define CALL(HH, KK) if (H == HH && K == KK && m <= XW_ODE_MAX_LAYERS) return xw_ode_fwd_multi_w##HH##_##KK(jobs, njobs, t, theta, method, L, d, m, zero16, stream);
  XW_ODE_WIDTHS(CALL)
#undef CALL
  if (H == XW_WIDE_H && K == XW_WIDE_K && m <= XW_ODE_MAX_LAYERS) return xw_ode_fwd_multi_w64_16(jobs, njobs, t, theta, method, L, d, m, zero16, stream);
  return xwg_ode_fwd_multi(jobs, njobs, t, theta, method, L, d, H, K, m, zero16, stream);    // any other width: the generic path
}

extern "C" int xw_ode_fwd(const double* xT, const double* t, const double* start, const double* theta, int method, int N,
                          int L, int d, int H, int K, int m, double* u, double* Y, void* stream) {
  XwOdeFwdJob j = {xT, start, u, Y, nullptr, N, 0, 0, 0};
  return xw_ode_fwd_multi(&j, 1, t, theta, method, L, d, H, K, m, nullptr, stream);
}

extern "C" int xw_ode_bwd_multi(const XwOdeBwdJob* jobs, int njobs, const double* t, const double* theta, int method, int L,
                                int d, int H, int K, int m, int mode, void* stream) {
#define CALL(HH, KK) if (H == HH && K == KK && m <= XW_ODE_MAX_LAYERS) return xw_ode_bwd_multi_w##HH##_##KK(jobs, njobs, t, theta, method, L, d, m, mode, stream);
  XW_ODE_WIDTHS(CALL)
#undef CALL
  if (H == XW_WIDE_H && K == XW_WIDE_K && m <= XW_ODE_MAX_LAYERS) return xw_ode_bwd_multi_w64_16(jobs, njobs, t, theta, method, L, d, m, mode, stream);
  return xwg_ode_bwd_multi(jobs, njobs, t, theta, method, L, d, H, K, m, mode, stream);
}

extern "C" int xw_ode_bwd(const double* xT, const double* t, const double* start, const double* theta, const double* Y,
                          const double* ubar, int method, int N, int L, int d, int H, int K, int m, int mode, double* gx,
                          double* gs, double* gslab, void* stream) {
  XwOdeBwdJob j = {xT, start, Y, nullptr, ubar, gx, gs, gslab, N, 0, nullptr, nullptr, 0.0, 0.0, 0, nullptr, nullptr, nullptr, 0.0};
  return xw_ode_bwd_multi(&j, 1, t, theta, method, L, d, H, K, m, mode, stream);
}
