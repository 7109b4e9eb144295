// xw_generic.h -- the generic-width path behind the public stepper / test-network entry points (xw_generic.hip).  Library-internal:
// hidden visibility, the shared object exports xw_* only.
#pragma once
#include "xnwan.h"
#define XWG_MAX_H 64     /* u_hidden_dim */
#define XWG_MAX_K 16     /* u_hidden_hidden_dim */
#define XWG_MAX_W 128    /* v_hidden_dim */
#define XWG_MAX_Q 16     /* v_layers */
#define XWG_MAX_M 32     /* u_layers (the MFMA containers stop at XW_ODE_MAX_LAYERS = 10; deeper fields run here, at their own widths) */
#define XWG_HIDDEN __attribute__((visibility("hidden")))
XWG_HIDDEN int xwg_ode_ok(int d, int H, int K, int m);
XWG_HIDDEN int xwg_disc_ok(int d, int W, int q);
XWG_HIDDEN int xwg_ode_fwd_multi(const XwOdeFwdJob* jobs, int njobs, const double* t, const double* theta, int method, int L, int d,
                                 int H, int K, int m, double* zero16, void* stream);
XWG_HIDDEN int xwg_ode_bwd_multi(const XwOdeBwdJob* jobs, int njobs, const double* t, const double* theta, int method, int L, int d,
                                 int H, int K, int m, int mode, void* stream);
XWG_HIDDEN int xwg_disc_fwd(const double* xT, const double* t, const double* tpp, const double* phi, int N, int L, int d, int W, int q,
                            double* v, double* vt, double* gxv, double* gtv, int ngrad, double* act, void* stream);
XWG_HIDDEN int xwg_disc_bwd(const double* xT, const double* t, const double* tpp, const double* phi, const double* vbar, int N, int L,
                            int d, int W, int q, const double* act, double* gslab, int nslab, void* stream);
