/* xnwan.h -- C ABI of libxnwan.so, the MI355X (gfx950) XNODE-WAN hot-path library.
 *
 * Every entry point replaces a piece of the reference's Python/ATen hot path (file:line are into
 * paulvoliva/XNODE-WAN-PDE-solver @ v1).  The reference has no FFI of its own: these are the calls a
 * maintainer would bind (ctypes stub in INTEGRATION.md) from src/model.py, src/loss.py and src/training.py.
 *
 * Conventions
 *   - all pointers are DEVICE pointers to contiguous arrays owned by the caller; nothing is allocated inside
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues work on it (graph-capture safe)
 *   - return value: 0 = ok, negative = argument/shape error (XW_E_*), positive = hipError_t of the launch
 *   - per-point arrays are TIME-MAJOR: a[l*N + n] for time index l and path n   ("[L,N]")
 *   - sample coordinates are passed transposed: xT[i*N + n] = x_n[i] ("[d,N]"), time grid t[L]; both float64 (the
 *     reference's cube samples are float32 and up-cast exactly at the first layer, src/model.py:46,154; its sphere
 *     samples are float64, src/dataset.py:65-96)
 *   - theta / phi are the parameter blobs of u_theta / v_phi in named_parameters() order (float64):
 *       theta: IL0.w[H,1] IL0.b[H] IL2.w[H,H] IL2.b[H] IL4.w[H,H] IL4.b[H]            (src/model.py:78)
 *              Win[K,d+1+H] (columns: x(d) | t | y(H)) Win.b[K] Wh[K,K] Wh.b[K] Wo[H,K] Wo.b[H]   (:130-138)
 *              FL.w[1,H] FL.b[1]                                                      (:85)
 *       phi:   Vin[W,d+1] (columns: t | x(d)) Vin.b[W] Vh[W,W] Vh.b[W] Vo[1,W] Vo.b[1]    (:34-36)
 *   - method: 0 = euler, 1 = midpoint, 2 = rk4 (3/8 rule)  -- fixed grid == the sample times (src/model.py:103-106)
 */
#ifndef XNWAN_H
#define XNWAN_H
#ifdef __cplusplus
extern "C" {
#endif

#define XW_E_DIMS     (-1) /* (H,K) or W not among the compiled instantiations */
#define XW_E_ARG      (-2) /* null pointer / non-positive size / bad enum */
#define XW_E_WORKSPACE (-3) /* workspace too small */
#define XW_E_COMM     (-4) /* RCCL not available / a collective call failed */

/* library identification; also lets the host check that the .so it loaded is this ABI */
int xw_abi_version(void);
/* writes the supported (H,K) pairs and W values as a NUL-terminated string into buf */
int xw_supported_dims(char* buf, int buflen);

/* sizes of the parameter blobs (doubles) */
int xw_theta_size(int d, int H, int K);
int xw_phi_size(int d, int W);

/* ---- u_theta: NeuralODE.forward for a group of N equal-length paths (src/model.py:87-112,140-156) -------------
 * start[N]: the scalar initial value h(x_n) or g(t_0,x_n) (src/model.py:95-96).
 * u[L,N] out; Y[L,H,N] out (hidden state at every sample time; needed by xw_ode_bwd), may be NULL. */
int xw_ode_fwd(const double* xT, const double* t, const double* start, const double* theta,
               int method, int N, int L, int d, int H, int K, int m,
               double* u, double* Y, void* stream);

/* The same for up to 4 independent groups of paths in ONE launch (interior + boundary sample, ...): one wave per 16
 * paths fills only a quarter of an MI355X at N = 4096, so independent groups are co-scheduled explicitly. */
/* act (may be NULL): activation store of (L-1) x xw_ode_act_rows() x (N rounded up to a multiple of 16) doubles (its
 * layout is the kernels' own: [step][tile of 16 paths][row][16]) -- the forward pass keeps the layer inputs of every
 * stage of every step so that the sweeps (XwOdeBwdJob.act) read them back instead of re-evaluating the field: the
 * record is 180 doubles (+ 4 of ReLU-mask words, all an x-only sweep reads besides the tanh rows) per path and step at (H, K, m) = (20, 10, 8) with midpoint, HBM capacity and bandwidth are idle
 * on this path, and the lone sweep wave saves 58 MFMAs + two tanh blocks per step.  Ignored by rk4. */
typedef struct { const double* xT; const double* start; double* u; double* Y; double* act; int N;
                 int act_x_only;   /* != 0: store only what a sweep WITHOUT weight gradients (mode 1) reads back -- the tanh
                                      rows and the ReLU-mask words, 1/7 of the bytes; same for every job of a launch */
                 int narrow;       /* != 0: NARROW TILES -- the 16 paths of a tile as four waves of 4 paths x 16 rows
                                      (csrc/xw_ode_n4.h; see xw_ode_bwd mode bit 4): the same outputs and the same activation
                                      store, for launches that leave SIMDs idle; same for every job of a launch */
                 int prio_drop;    /* 0..3: the launch's waves run at wave priority 3 - prio_drop (job 0 decides).  The stepper's
                                      chains share their SIMDs with the throughput-bound test network and run at raised priority
                                      by default; a launch that is NOT on the sub-step's critical path gives that up */
               } XwOdeFwdJob;
/* rows of the activation record per step (0: this method's sweeps recompute; negative: XW_E_*) */
int xw_ode_act_rows(int method, int H, int K, int m);
/* zero16 (may be NULL): 16 doubles cleared by the launch -- the sub-step's partial-sum slots scal[], so that no separate
 * memset sits at the head of the critical path */
int xw_ode_fwd_multi(const XwOdeFwdJob* jobs, int njobs, const double* t, const double* theta,
                     int method, int L, int d, int H, int K, int m, double* zero16, void* stream);

/* number of partial-gradient slabs xw_ode_bwd writes for N paths, and doubles of workspace it needs */
int xw_ode_bwd_slabs(int N);

/* Reverse sweep through the discrete stepper (autograd replacement for src/loss.py:55 and src/training.py:137).
 * ubar[L,N]: cotangent on u (NULL = all ones).
 * mode bit 0: produce gx[d,N] = d<ubar,u>/dx_n  and gs[N] = d<ubar,u>/d start_n     (nabla_x u of src/loss.py:56-58)
 * mode bit 1: produce parameter-gradient slabs gslab[xw_ode_bwd_slabs(N)][P_u] (to be summed by xw_adam / xw_slab_sum)
 * mode bit 2 (only with bits 0 and 1): the caller guarantees ubar == 1 at every time index >= 1; gx, gs are then
 *   returned for the ALL-ONES cotangent (the helper backward of src/loss.py:55) while the parameter gradients use ubar
 *   itself -- the pollution sweep and the nabla_x u sweep of a generator sub-step are the same adjoint, run once.
 *   In the multi-group form jobs with gx == gs == NULL simply produce no x outputs.
 * mode bit 3 (not with bit 2): adjoint = True of the reference (src/model.py:103, torchdiffeq.odeint_adjoint, 0.1.1):
 *   the continuous adjoint instead of the reverse of the steps taken.  For i = L-1 .. 1 the augmented state
 *   (y, a, theta-bar) is restarted from the checkpoint y(t_i) and advanced by ONE step of `method` from t_i to t_{i-1}
 *   under  d/dt (y, a, theta-bar) = (f, -a^T df/dy, -a^T df/dtheta);  then a += flw * ubar[i-1].  The sample point x is
 *   not an input of that adjoint: gx is returned as zero, gs (through the lift) and the parameter gradients as usual.
 *   Differs from the discrete sweep by O(dt^p); the activation store is not used.  torchdiffeq is absent from the
 *   reference tree: restated from its published algorithm, parity unpinned.
 * mode bit 4 (euler / midpoint with an activation store, not with bit 3): NARROW TILES -- the same sweep with the 16 paths of
 *   a tile spread over four waves of 4 paths x 16 rows (csrc/xw_ode_n4.h): four times the instruction streams, each a
 *   shorter dependent chain, ~1.8 x the matrix-pipe time per path.  Same inputs, outputs and slab count; results differ from
 *   the 16-path form only by the summation order of the weight gradients.  For launches that leave SIMDs idle (fewer
 *   16-path tiles than the chip has SIMDs and nothing else running beside them).
 * mode bits 5..6: priority drop 0..3 -- the launch's waves run at wave priority 3 - drop (see XwOdeFwdJob.prio_drop): for sweeps
 *   that are not on the sub-step's critical path (the generator's sweeps A + boundary: 0.5027 -> 0.4955 ms per sub-step). */
int xw_ode_bwd(const double* xT, const double* t, const double* start, const double* theta, const double* Y,
               const double* ubar, int method, int N, int L, int d, int H, int K, int m, int mode,
               double* gx, double* gs, double* gslab, void* stream);

/* multi-group form: every job has its own sample, checkpoints, cotangent and outputs; `mode` is common to all jobs */
typedef struct { const double* xT; const double* start; const double* Y; const double* act; const double* ubar;
                 double* gx; double* gs; double* gslab; int N;
                 /* cotangent formed from a residual on the fly instead of a stored one (res_u != NULL; ubar must then be
                  * NULL):  ubar[l][n] = res_base + res_coef (res_u[l][n] - ref),  with ref = res_ref[l][n] at every time
                  * index (res_first_only == 0: the boundary penalty of src/loss.py:84, res_u = u on the boundary paths,
                  * res_ref = g) or ref = res_ref[n] and only at l = 0 (res_first_only != 0: the initial-value penalty of
                  * src/loss.py:79 on top of the constant res_base).  The sweeps that only need these penalties start right
                  * behind the forward pass, without a cotangent kernel in between. */
                 int res_first_only; const double* res_u; const double* res_ref; double res_coef; double res_base;
                 /* res_first_only == 2: the weak form's dI/du (basis B of xw_gen_cotangents, src/loss.py:64,70) --
                  *   ubar[l][n] = res_coef d(c(u) u)/du v w  (+ res_base v at l = L-1),   res_u = u, res_ref = v,
                  *   w = res_w[n] or res_w[l][n] (res_w_per_point), d(c u)/du = res_c + u res_cp (tabulated c, dc/du) or
                  *   res_kappa2 u (c = kappa u, res_c == NULL): sweep B starts right behind the test network. */
                 int res_w_per_point; const double* res_w; const double* res_c; const double* res_cp; double res_kappa2;
               } XwOdeBwdJob;
int xw_ode_bwd_multi(const XwOdeBwdJob* jobs, int njobs, const double* t, const double* theta,
                     int method, int L, int d, int H, int K, int m, int mode, void* stream);

/* ---- v_phi: discriminator.forward (src/model.py:37-47) + d/dt by forward-mode ------------------------------------
 * Path mode (tpp == NULL): point (l,n) = (t[l], x_n).  Point mode (tpp != NULL): L must be 1, point n = (tpp[n], x_n).
 * v[L,N] out; vt[L,N] out = dv/dt (may be NULL).
 * gxv[d,ngrad], gtv[ngrad] (may be NULL): input gradient of v (nabla_x v, dv/dt by reverse mode) for the LEADING ngrad
 * points in time-major order -- the weak form reads nabla phi only at the first time index (pass ngrad = N), so this
 * fuses what would otherwise be a separate xw_disc_gradx launch; needs q <= 16.
 * max_blocks: cap on the grid (0 = default, two resident blocks per CU); a smaller grid leaves SIMD slots to kernels
 * that run concurrently. */
int xw_disc_fwd(const double* xT, const double* t, const double* tpp, const double* phi,
                int N, int L, int d, int W, int q, double* v, double* vt, double* gxv, double* gtv, int ngrad,
                int max_blocks, double* act, void* stream);
/* The same with the x-projection of the input layer HOISTED out of the points (path mode only, tpp == NULL): on vertical paths the d
 * spatial columns of the input layer do not move along a path, so Vin[:, 1..d] x_n + Vin.b is formed once per path by xw_disc_xproj
 * (xproj[16 MT][N]: 64 rows for the widths 50 and 64, 96 / 128 for the wide containers; rows >= W zero) and a point's input layer is one load and one multiply-add per row -- instead of ceil(d/4) x 4
 * matrix instructions and as many loads of x per 16-point tile, L times per path (12 % of the launch at d = 100).  xproj == NULL: as
 * xw_disc_fwd.  The widths of the MFMA kernels only (50, 64, 96, 128). */
int xw_disc_xproj(const double* xT, const double* phi, int N, int d, int W, double* xproj, void* stream);
int xw_disc_fwd_xproj(const double* xT, const double* t, const double* tpp, const double* phi,
                      int N, int L, int d, int W, int q, double* v, double* vt, double* gxv, double* gtv, int ngrad,
                      int max_blocks, double* act, const double* xproj, void* stream);
/* act (may be NULL): activation record of xw_disc_act_rows(W, q) x (N L rounded up to a multiple of 16) doubles = the
 * inputs relu(a_j) of the q tied layers and tanh(a_q).  Its layout is the kernels' own (tile-major: [tile of 16 points]
 * [row j W + k][16], so that a wave's accesses are contiguous); callers only size it.  Given to xw_disc_bwd it replaces that kernel's forward recompute (500 doubles per point =
 * 524 MB at the headline size; written at ~2 TB/s next to a matrix-bound kernel, read once by the backward). */
int xw_disc_act_rows(int W, int q);

/* input gradient of <vbar, v> at a set of points (reverse mode, no parameter gradients): gxv[d,N] (nabla_x) and
 * gtv[N] (d/dt), for the N points (t_n, x_n) with t_n = tpp ? tpp[n] : t[0]; vbar[N] or NULL (= ones).
 * (XV.grad of src/loss.py:60-63; the fused step only needs it at the first time index)
 * Compiled for the reference's width and depth, W = 50, q = 9; other depths and W = 64, 96, 128 (XW_E_DIMS here) take the same
 * gradient from xw_disc_fwd's gxv/gtv outputs, which run at any q <= 16, and scale it by vbar. */
int xw_disc_gradx(const double* xT, const double* t, const double* tpp, const double* phi, const double* vbar,
                  int N, int d, int W, int q, double* gxv, double* gtv, void* stream);

int xw_disc_bwd_slabs(int N, int L);
/* parameter gradient of <vbar, v>: slabs gslab[xw_disc_bwd_slabs][P_v] (input gradient: xw_disc_gradx).
 * act: the record xw_disc_fwd stored for the same phi and points, or NULL (the forward is then recomputed per tile).
 * Any depth q >= 0 and all four widths (W = 50, 64, 96, 128) run from the record; the recomputing form keeps its checkpoints in
 * registers and is compiled for the reference's W = 50, q = 9 only (XW_E_DIMS otherwise). */
int xw_disc_bwd(const double* xT, const double* t, const double* tpp, const double* phi, const double* vbar,
                int N, int L, int d, int W, int q, const double* act, double* gslab, void* stream);

/* ---- weak functional and cotangents (src/loss.py:46-96) -----------------------------------------------------------
 * scal[16] (device, float64): 0 I   1 S=sum v^2   2 SSE_init   3 SSE_bdry   4 loss_u   5 loss_v   6 int
 *                             7 sum u   8 sum d(phi)/dt  (pairwise groups only, folded into I by xw_pair_fold)   (rest reserved)
 * xw_weak_partials ADDS this rank's partial sums into scal[0..2] (and [7..8]); zero scal first; all-reduce scal[0..8]
 * across ranks.
 *   The grid-wide sums are deterministic (per-block partials + last-block final sum, no float atomics): `work` is a
 *   caller-provided scratch of xw_reduce_work_size() doubles, zero-initialised ONCE (the kernels leave it clean), not
 *   shared between launches that may run concurrently.
 *   w: distance-to-boundary weight, per path (w_per_point=0, [N]) or per point ([L,N]);  wt: d w/dt [L,N] or NULL (=0)
 *   s3x[N]: the l=0 gradient-contraction term  sum_ij a_ij d_i phi d_j u + sum_i b_i phi d_i u  (src/loss.py:66-69);
 *           NULL for a = identity, b = 0: then it is contracted in-kernel from gx[d,N], gs[N] (xw_ode_bwd), ghT[d,N]
 *           (nabla_x of the start value), gxv[d,N] (xw_disc_gradx), w0[N], gwx0T[d,N] (w and nabla_x w at t_0)
 *   c, cp: c(u,t,x) and dc/du, [L,N]; both NULL means c = ckappa * u           f[L,N]; h[N]
 *   pairwise != 0 (needs L == 1): the single-slice group at T0 of a list domain, where NeuralODE.forward returns [N,1]
 *   instead of [N,1,1] (src/model.py:89-91) and src/loss.py:65,70 broadcast [N] against [N,1] into [N,N] tables over all
 *   PAIRS of paths, summed over both axes.  The sums factorise (O(N)): the d(phi)/dt term becomes (sum u)(sum dphi/dt) --
 *   the factors go to scal[7], scal[8] and are folded into scal[0] by the finalisation (or by xw_pair_fold after the
 *   all-reduce) -- and  sum_mn (s31_m + c_n u_n phi_n + f_m phi_n) = N sum_n (s31_n + c_n u_n phi_n + mean(f) phi_n):
 *   the caller passes s3_scale = Nglob and f := mean f in every entry.  href[N] (may be NULL = h): reference of the
 *   initial penalty, := mean h for the pairwise mean of src/loss.py:79, whose sample-only rest var(h) arrives as init_off
 *   (bdry_off likewise for a single-slice T0 boundary group, src/loss.py:84).  s3_scale = 1, offsets 0 otherwise.
 *   Vol = domain volume; Nglob = global number of interior paths (the 1/N, 1/(N L) factors of src/loss.py:64-71)
 *   finalize != 0 (single-GPU path): the sums of this launch are the global ones, so the block that completes them also
 *   does what xw_losses does (loss values into scal[4..6] from scal[0..3], optimiser counter *step += 1 if step != NULL);
 *   Lb, Nbglob, alpha as in xw_losses.  With several GPUs call xw_losses after the all-reduce instead.
 *   ub, gb [Pb] (or NULL, NULL, 0): the boundary forward u_b and the boundary data g at the Pb = N_b L_b boundary points -- the
 *   launch then also adds sum (u_b - g)^2 to scal[3] (what xw_bdry_partials does in a launch of its own, src/loss.py:84). */
int xw_weak_partials(const double* u, const double* v, const double* vt, const double* w, int w_per_point,
                     const double* wt, const double* s3x, const double* gx, const double* gs, const double* ghT,
                     const double* gxv, const double* w0, const double* gwx0T, int d, const double* c, double ckappa,
                     const double* f, const double* h, const double* href, int pairwise, double s3_scale, int N, int L,
                     double Vol, double Nglob, double* work, double* scal, int finalize, int Lb, double Nbglob, double alpha,
                     double init_off, double bdry_off, long long* step, const double* ub, const double* gb, long Pb,
                     void* stream);
/* scal[0] -= (Vol / Nglob) scal[7] scal[8]; scal[7] = scal[8] = 0: the pairwise d(phi)/dt term of a single-slice T0 group,
 * once its two factors are global (several GPUs: after the all-reduce; one GPU: done by xw_weak_partials' finalisation) */
int xw_pair_fold(double* scal, double Vol, double Nglob, void* stream);
int xw_reduce_work_size(void);
/* Distance weight of the hypercube domain and its x-gradient at N points (replaces Hypercube.func_w, src/dataset.py:278-282, and
 * the autograd pass through it that the weak form takes, src/loss.py:51-63), in the sample's float32 arithmetic like the
 * reference's tensors, widened to float64 on the way out:
 *   x[N, d] float32 row-major;  w[n] = min(min_i |top - x_i|, min_i |bot - x_i|)  (also into w0 if not NULL);
 *   gwT[d, N] = d w / d x as autograd returns it (first minimal index; an exact tie of the two distances splits half / half);
 *   xT[d, N] (or NULL) = the points, transposed.  One launch where the tensor formulation takes ~25. */
int xw_cube_weight(const float* x, int N, int d, double top, double bot, double* w, double* w0, double* gwT, double* xT,
                   void* stream);
/* The l = 0 gradient-contraction term for GENERAL coefficients (src/loss.py:66-69 with the a[d,d,N,L] / b[d,N,L] tables of
 * src/training.py:32-41 -- of which only time index 0 can ever contribute, so only that slice is tabulated):
 *   s3x[n] = sum_ij a_ij(t_0, x_n) d_i phi d_j u + phi sum_i b_i(t_0, x_n) d_i u,
 *   nabla phi = w0 gxv + v0 gwx0T,   nabla u = gx + gs ghT,   phi = v0 w0          (all [d,N] / [N] as in xw_weak_partials)
 * amode: 0 a = identity (A0 ignored), 1 A0[d,d] constant matrix, 2 A0[d,N] diagonal, 3 A0[d,d,N] full table.
 * B0[d,N] or NULL (b = 0).  The result feeds xw_weak_partials(s3x = ...). */
int xw_weak_contract_general(const double* A0, int amode, const double* B0, const double* gx, const double* gs,
                             const double* ghT, const double* gxv, const double* w0, const double* gwx0T, const double* v0,
                             int d, int N, double* s3x, void* stream);
/* boundary penalty partial: scal[3] += sum (u_b - g)^2 ; ubar_b = alpha * 2 (u_b - g) / (Nbglob * L) */
int xw_bdry_partials(const double* ub, const double* g, int Nb, int L, double alpha, double Nbglob,
                     double* ubar_b, double* work, double* scal, void* stream);
/* generator cotangent bases on u (loss_u of src/loss.py:93 + the pollution of :55).  Both are available right after
 * the forward passes (neither needs the global I), so the parameter sweeps need not wait for the x-sweep:
 *   ubarA = pollution + alpha * 2 (u[0,n] - h_n) / Nglob at l = 0
 *   ubarB = dI/du = (V/N) v[L-1,n] at l = L-1  +  (V/(N L)) (c + u dc/du) v w
 *   d loss_u / d theta = J^T ubarA + (2 / I) J^T ubarB + (boundary sweep);  the 2/I is applied by xw_adam.
 * Either output may be NULL; basis A does not read v (it can be formed before the test network has run).
 * scal != NULL (merged form, ubarB must be NULL): the global I = scal[0] is already known, ubarA := A + (2/I) B --
 * one interior sweep instead of two (single-GPU path; the split form buys the single all-reduce on several GPUs). */
int xw_gen_cotangents(const double* u, const double* v, const double* w, int w_per_point, const double* c,
                      const double* cp, double ckappa, const double* h, int N, int L, double Vol, double Nglob,
                      double alpha, double pollution, const double* scal, double* ubarA, double* ubarB, void* stream);
/* discriminator cotangent on v (loss_v of src/loss.py:96 + the pollution of :60); reads I = scal_in[0], S = scal_in[1];
 * s3_scale: 1, or Nglob on a pairwise group (with f := mean f), see xw_weak_partials */
int xw_disc_cotangent(const double* u, const double* v, const double* w, int w_per_point, const double* c,
                      double ckappa, const double* f, const double* h, int N, int L, double Vol, double Nglob,
                      double pollution, double s3_scale, const double* scal_in, double* vbar, void* stream);
/* scal[4] = loss_u, scal[5] = loss_v, scal[6] = int from the (all-reduced) partial sums scal[0..3] (src/loss.py:87-96);
 * L / Lb: sample times of the interior / boundary paths of the group;
 * step (may be NULL): optimiser step counter to increment here when xw_adam was called with bump_step = 0 */
int xw_losses(double* scal, int L, int Lb, double Vol, double Nglob, double Nbglob, double alpha, double init_off,
              double bdry_off, long long* step, void* stream);

/* ---- optimiser (torch.optim.Adam defaults, src/training.py:103-104) ------------------------------------------------
 * grad = gextraA + sum_s gslabA[s] + coefB * (gextraB + sum_s gslabB[s]),  coefB = scal ? 2 / scal[0] : 1
 * (slab sets [n][P]; gextra*[P] pre-reduced gradients, e.g. after an all-reduce; any of them may be NULL / 0)
 * state: m[P], v[P], step (device int64).  bump_step = 1: incremented here after the update; 0: left alone (a later
 * xw_losses does it); -1: xw_losses already advanced it for this update (lets Adam be the last kernel of a sub-step)
 * [lag_lo, lag_hi): parameters with their own step count *step - *lag (lag: device int64, may be NULL with an empty
 * range).  skip != 0: that range is left untouched (no moment decay, no step) and *lag += 1 -- torch.optim.Adam skips a
 * parameter whose .grad is None, which is the state of u_theta's field parameters while no group of a sub-iteration has
 * integrated the ODE (single-slice groups of the list domains, src/model.py:89-91 + src/training.py:127,138). */
int xw_adam(double* param, const double* gslabA, int nA, const double* gextraA, const double* gslabB, int nB,
            const double* gextraB, const double* scal, double* m, double* v, long long* step, int bump_step, int P,
            double lr, double beta1, double beta2, double eps, double* gsum_out, int lag_lo, int lag_hi, int skip,
            long long* lag, void* stream);
/* plain slab reduction: out[P] = (accumulate ? out : 0) + sum_s gslab[s][P] */
int xw_slab_sum(const double* gslab, int nslab, int P, int accumulate, double* out, void* stream);
/* two slab sets of the same P in one launch: outA[P] = sum of gA[nA][P], outB[P] = sum of gB[nB][P] (the generator
 * sub-step's exchange buffer on several GPUs) */
int xw_slab_sum2(const double* gA, int nA, double* outA, const double* gB, int nB, double* outB, int P, void* stream);

/* ---- one optimiser sub-step of ONE group of paths as ONE call (src/training.py:127-138 generator, :152-162 discriminator) ----
 * The loops `for (datau, datav, bdata) in points:` of the reference visit 11-20 groups per sub-iteration on the time-varying
 * domains, every group with new shapes every sample: nothing to replay, and a host language that issues the ~15 launches
 * of a group sub-step one by one spends more time issuing than the GPU spends running them.  These two entry points enqueue
 * the whole chain on ONE stream (program order is the dependency), from two plain structs the caller fills when a group is
 * loaded: the kernels, their arguments and their order are exactly those of the separate calls above.
 * Scope: c(u,t,x) = ckappa u or tabulated c / cp already in the group (the caller refreshes them), any a / b (A0 / B0 tables),
 * pairwise single-slice groups, carried gradients of a multi-group sub-iteration; one process, or -- XwGroup.sharded with
 * XwSolverState.exchange -- one rank of several that hold contiguous shares of the group's paths: the same chain with the
 * exchange steps of dist.py between its launches (generator: ONE sum of pack_u = [J^T ubarA | J^T ubarB | scal] over the ranks;
 * discriminator: scal[0..8], then the packed gradient grad_v), every rank applying the identical update.  A share may be EMPTY
 * (N == 0 and / or Nb == 0: a group with fewer paths than ranks): nothing is launched for what the rank does not hold, zeros go
 * into every exchange, the update is the same (nn.DataParallel scatters a batch of any size, src/training.py:93-97). */
typedef struct {
  int N, Nb, L, Lb, d;            /* interior paths, boundary paths, their sample times, dimensions */
  int same_grid;                  /* boundary paths on the interior's time grid (one launch for both) */
  int w_per_point, amode, pair_i, pair_b;
  int ns_u, ns_b;                 /* slabs of the interior / boundary sweeps (xw_ode_bwd_slabs) */
  int narrow;                     /* narrow-tile launches: bit 0 forward (generator), 1 boundary forward alone, 2 the sweeps launch of the
                                     generator sub-step (A, boundary on the same grid, B: one launch), 3 boundary sweep alone, 4 unused,
                                     5 x-only sweep (generator, unfused), 6 forward (discriminator), 7 x-only sweep (discriminator) */
  int sharded;                    /* != 0: N / Nb are this rank's SHARE of a group of Nglob / Nbglob paths (either may be 0); the
                                     sub-step runs XwSolverState.exchange between its launches */
  double Vol, Nglob, Nbglob, s3_scale, init_off, bdry_off, ckappa;
  const double *xT, *xvT, *xbT, *t, *tb, *tpp, *xvT_pts;
  const double *start, *start_b, *h, *href, *f, *g, *w, *wt, *w0, *ghT, *gwx0T, *c, *cp, *A0, *B0;
  double *u, *ub, *Y, *Yb, *act, *act_b, *v, *vt, *gxv, *gtv, *gx, *gs, *vbar, *s3x, *vact, *slabA, *slabB, *slab_v, *work_i, *work_b;
  double *xproj;                  /* [64][N] ([96][N] / [128][N] at W = 96 / 128) or NULL: table of xw_disc_xproj -- the test network then runs as xw_disc_fwd_xproj (path mode) */
} XwGroup;
/* in-place float64 sum of buf[count] over the ranks, enqueued on `stream`: xw_allreduce's own signature */
typedef int (*XwExchangeFn)(double* buf, int count, void* ctx, void* stream);
typedef struct {
  int method, H, K, m, W, q, Pu, Pv, adjoint;
  int v_blocks, v_blocks_disc;    /* grid caps of the test network's launch in the two sub-steps (0 = default) */
  int lag_lo, lag_hi;             /* xw_adam: the field's range of theta keeps its own step count */
  double alpha, pollution, lr_u, lr_v, beta1, beta2, eps;
  double *theta, *phi, *scal, *grad_u, *grad_v, *m_u, *v_u, *m_v, *v_v;
  long long *step_u, *step_v, *lag_u;
  /* several ranks (NULL / unused in one process): the in-place float64 sum over the ranks -- xw_allreduce itself with
   * exchange_ctx = the communicator, or a host-side stand-in with the same contract (rehearsals without RCCL) -- and the
   * generator sub-step's exchange buffer pack_u[2 Pu + 16] = [sum A | sum B | scal], i.e. scal == pack_u + 2 Pu */
  XwExchangeFn exchange;
  void* exchange_ctx;
  double* pack_u;
} XwSolverState;
/* skip_v: v, dv/dt, nabla_x v(t_0) of this group are still those of the current phi and sample (opt-in reuse);
 * store_record: the test network's forward also stores its layer inputs (vact) for xw_disc_bwd;
 * accum (may be NULL): gradient carried from the earlier groups of the sub-iteration (added in, then overwritten with the sum);
 * adam_skip_field: no group of this sub-iteration has integrated the ODE yet (xw_adam skip) */
int xw_substep_gen(const XwGroup* g, const XwSolverState* s, int skip_v, int store_record, double* accum, int adam_skip_field,
                   void* stream);
int xw_substep_disc(const XwGroup* g, const XwSolverState* s, int skip_v, int use_record, double* accum, void* stream);

/* ---- the exchange step of the sharded path (replaces nn.DataParallel's scatter / gather / reduce-add around both nets,
 * src/training.py:93-97).  One process per GPU; every rank owns a contiguous share of the Monte-Carlo paths; per
 * generator sub-step ONE packed buffer [J^T ubarA | J^T ubarB | partial sums] is summed over the ranks, per
 * discriminator sub-step the partial sums (I, sum v^2) and then the packed gradient (dist.py).  RCCL over xGMI, bound
 * with dlopen (the library loads without it; these four calls then return XW_E_COMM).
 *   xw_comm_available : 0 if RCCL could be bound in this process, XW_E_COMM if not; NOT collective -- the ranks agree on
 *                       it before any of them enters the collective xw_comm_init
 *   xw_comm_unique_id : rank 0 creates the 128-byte rendezvous id; the host side carries it to the other ranks
 *   xw_comm_init      : collective over all ranks, on the caller's current HIP device; *comm is an opaque handle
 *   xw_allreduce      : in-place float64 sum of buf[count] over the ranks; only enqueues on `stream` (graph-capture
 *                       safe: a sub-step and its exchange replay as one HIP graph)
 *   xw_comm_destroy   : releases the handle */
int xw_comm_available(void);
int xw_comm_unique_id(unsigned char* id128);
int xw_comm_init(const unsigned char* id128, int nranks, int rank, void** comm);
int xw_allreduce(double* buf, int count, void* comm, void* stream);
int xw_comm_destroy(void* comm);

/* ---- sample fields of a list-domain sample (callers' side of the path: src/training.py:13-43 func_eval + the per-group
 * tensor plumbing of src/loss.py:36-60) ----------------------------------------------------------------------------------
 * One row per output array: dst[i][j][k] = src[i*s0 + j*s1 + k*s2] for i < n0, j < n1, k < n2 (dst contiguous, float64);
 * `before` = number of elements of all rows in front of this one (the rows partition [0, total)).  The table lives in
 * device memory; count <= 1024. */
typedef struct XwGather {
  const double* src;
  double* dst;
  long n0, n1, n2;
  long s0, s1, s2;
  long before;
} XwGather;
int xw_gather_fields(const XwGather* table_dev, int count, long total, void* stream);

/* ---- host-side helper of the samplers (no GPU involved) ---------------------------------------------------------------
 * float32 uniform fill straight from torch's CPU generator state (torch.get_rng_state(), 5056 bytes: at::mt19937, legacy state
 * layout): the numbers, their order and the state left behind are those of Tensor.uniform_(from, to) on the default
 * generator (src/dataset.py:248-272 draws every sample that way: "same seeds" = this stream), 8 x faster than the scalar
 * walk.  fused != 0: the final scale-and-shift is one fma (what torch's build does on x86-64; the caller verifies against
 * torch once and falls back to torch otherwise).  Returns 0, or XW_E_ARG for a blob that is not such a state. */
int xw_mt19937_uniform_f32(void* state_blob, long blob_bytes, float* out, long n, float from, float to, int fused);

/* float64 standard normals from numpy's GLOBAL legacy generator state (np.random.get_state(): key[624], pos, has_gauss,
 * cached_gaussian -- the four are updated in place and go back with np.random.set_state): the numbers, their order and the
 * state left behind are those of np.random.normal(size = n) (legacy_gauss, polar method; the ball domains draw their points
 * that way, src/dataset.py:65, 180: "same seeds" = this stream), 2.3 x faster than numpy's value-by-value walk.  The caller
 * verifies against numpy once and falls back to numpy otherwise.  Returns 0, or XW_E_ARG for pos outside 0..624 / null pointers. */
int xw_mt19937_legacy_normal_f64(unsigned int* key, int* pos, int* has_gauss, double* cached_gaussian, double* out, long n);

#ifdef __cplusplus
}
#endif
#endif
