"""Tensor-level wrappers around the C ABI: every operand is validated on the host (device, dtype, contiguity,
shape) BEFORE a kernel is enqueued, then passed as a raw device pointer together with torch's current stream.

Conventions (include/xnwan.h): point arrays are time-major [L, N]; coordinates are transposed xT[d, N]; all float64.
"""
import torch

from ._lib import lib, check, XnwanError, XwOdeFwdJob, XwOdeBwdJob

METHODS = {'euler': 0, 'midpoint': 1, 'rk4': 2}
F32, F64 = torch.float32, torch.float64


def _need_gpu():
    if not torch.cuda.is_available():
        raise XnwanError('no GPU visible: the XNODE-WAN kernels only run on an MI355X (gfx950); there is no CPU path')


# Operand validation costs ~0.7 us per check and a list-domain outer iteration makes ~28 000 of them (1.9 ms of its ~23).  The
# engine's own buffers are allocated by the engine with exactly these shapes: while TRUSTED is set (engine.Engine sets it
# around the eager sub-steps of list-domain groups, whose launches are issued from Python one by one) the checks are skipped.
# Everything reachable from user code -- the module call surface, the custom operators, the first launch of every captured
# graph -- stays checked.
TRUSTED = False
LONE_NARROW = __import__('os').environ.get('XW_LONE_NARROW', '1') == '1'     # (ode_fwd below)


def _chk(t, dtype, shape, name):
    if t is None or TRUSTED:
        return
    if not (torch.is_tensor(t) and t.is_cuda):
        raise XnwanError('%s must be a CUDA/HIP tensor' % name)
    if t.dtype != dtype:
        raise XnwanError('%s must be %s, got %s' % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise XnwanError('%s must be contiguous' % name)
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise XnwanError('%s must have shape %s, got %s' % (name, tuple(shape), tuple(t.shape)))


def _p(t):
    return 0 if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream():
    """raw handle of torch's current stream on the current device (the C call: no Stream object per launch -- a list
    domain issues ~500 launches per outer iteration from Python)"""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def method_id(name):
    if name not in METHODS:
        raise XnwanError("solver %r is not a fixed-grid scheme of this engine (have: %s)" % (name, sorted(METHODS)))
    return METHODS[name]


# Kernel instantiations (csrc/xw_ode.hip XW_ODE_DISPATCH, csrc/xw_disc.hip): widths the MFMA kernels are compiled for.  Any
# smaller network runs EXACTLY inside the next larger instantiation: its parameters are embedded in a zero-padded blob
# (nets.Blob) -- padding units have zero in- and outgoing weights and zero bias, so they stay identically zero through
# relu / tanh, contribute exact zeros to every sum, and receive exactly zero gradients (Adam leaves them at zero).
# Anything wider runs, at its own widths, on the GENERIC path (csrc/xw_generic.hip: per-path / per-point code on the vector ALU,
# two to three orders of magnitude slower -- there so that every legal configuration of the reference trains).
ODE_WIDTHS = [(20, 10), (32, 12), (64, 16)]   # (u_hidden_dim, u_hidden_hidden_dim) containers, smallest first.  (64, 16) (round 6) is the
                                           # WIDE container: the field on 16x16x4 matrix instructions, one wave per tile (+ a partner in the sweep with weight gradients), depths 1..10
                                           # (deeper fields: the generic path, at the network's own widths)
DISC_WIDTHS = [50, 64, 96, 128]            # v_hidden_dim containers (W = 50: 3 MFMA row tiles + a 2-row vector tail; 64: 4 tiles;
                                           # 96, 128 (round 6): 6 / 8 tiles, one block per CU, forward + reverse from the record)
GENERIC_ODE_MAX = (64, 16)                 # csrc/xw_generic.h
GENERIC_DISC_MAX = 128


ODE_WIDE_MAX_DEPTH = 10                    # (64, 16): 4 (u_layers - 1) ReLU-mask bits per stage, two words at depth 10
ODE_MAX_DEPTH = 10                         # the other containers (csrc/xw_common.h XW_ODE_MAX_LAYERS); deeper fields, up to
GENERIC_ODE_MAX_DEPTH = 32                 # this (csrc/xw_generic.h XWG_MAX_M), run on the generic path at their own widths


def ode_container(H, K, m=1):
    """the widths the stepper kernels run a (u_hidden_dim H, u_hidden_hidden_dim K) network of u_layers = m at"""
    for Hc, Kc in ODE_WIDTHS:
        if m > (ODE_WIDE_MAX_DEPTH if (Hc, Kc) == ODE_WIDTHS[-1] else ODE_MAX_DEPTH):
            continue                       # (u_layers > 10: the generic path, at the network's own widths)
        if H <= Hc and K <= Kc and lib.xw_theta_size(1, Hc, Kc) > 0 and lib.xw_ode_act_rows(0, Hc, Kc, 1) >= 0:
            return Hc, Kc
    if H <= GENERIC_ODE_MAX[0] and K <= GENERIC_ODE_MAX[1] and m <= GENERIC_ODE_MAX_DEPTH and lib.xw_ode_act_rows(0, H, K, m) >= 0:
        return H, K                        # the generic path, at the network's own widths
    raise XnwanError('u_hidden_dim = %d, u_hidden_hidden_dim = %d, u_layers = %d: the stepper kernels serve widths up to %s (MFMA) / %s (generic '
                     'path), depths up to %d / %d' % (H, K, m, ODE_WIDTHS[-1], GENERIC_ODE_MAX, ODE_MAX_DEPTH, GENERIC_ODE_MAX_DEPTH))


def ode_generic(H, K, m=1):
    """True when (H, K) at u_layers = m is served by the generic (slow) path rather than by an MFMA instantiation"""
    return (H, K) not in ODE_WIDTHS or m > (ODE_WIDE_MAX_DEPTH if (H, K) == ODE_WIDTHS[-1] else ODE_MAX_DEPTH)


def disc_container(W):
    for Wc in DISC_WIDTHS:
        if W <= Wc and lib.xw_disc_act_rows(Wc, 1) >= 0:
            return Wc
    if W <= GENERIC_DISC_MAX and lib.xw_disc_act_rows(W, 1) >= 0:
        return W
    raise XnwanError('v_hidden_dim = %d: the test-network kernels serve widths up to %d (MFMA) / %d (generic path)'
                     % (W, DISC_WIDTHS[-1], GENERIC_DISC_MAX))


def disc_generic(W):
    return W not in DISC_WIDTHS


def theta_size(d, H, K):
    return lib.xw_theta_size(d, H, K)


def phi_size(d, W):
    return lib.xw_phi_size(d, W)


def ode_fwd(xT, t, start, theta, method, H, K, m, want_Y=True, u=None, Y=None):
    """u_theta on N paths: returns (u[L,N], Y[L,H,N] or None)."""
    _need_gpu()
    d, N = xT.shape
    L = t.shape[0]
    _chk(xT, F64, (d, N), 'xT'); _chk(t, F64, (L,), 't'); _chk(start, F64, (N,), 'start')
    _chk(theta, F64, (theta_size(d, H, K),), 'theta')
    if u is None:
        u = torch.empty(L, N, dtype=F64, device=xT.device)
    if Y is None and want_Y:
        Y = torch.empty(L, H, N, dtype=F64, device=xT.device)
    _chk(u, F64, (L, N), 'u'); _chk(Y, F64, (L, H, N), 'Y')
    if Y is None and LONE_NARROW and (N + 15) // 16 <= 256 and (H, K) in ODE_WIDTHS:
        # an evaluation on its own (the diagnostic, a stop hook, Engine.predict) has the chip to itself: up to 256 tiles the narrow
        # layout -- four waves of 4 paths per tile, every SIMD busy -- ends sooner (57 against 73 us at 4096 paths x 32 times,
        # 109 against 143 at 64 times; beyond 256 tiles it loses: tools/lone_forward.py).  Same values to the last bit or two.
        arr = (XwOdeFwdJob * 1)()
        a = arr[0]
        a.xT, a.start, a.u, a.Y, a.act, a.N, a.act_x_only, a.narrow, a.prio_drop = _p(xT), _p(start), _p(u), 0, 0, N, 0, 1, 0
        check(lib.xw_ode_fwd_multi(arr, 1, _p(t), _p(theta), method, L, d, H, K, m, 0, _stream()), 'xw_ode_fwd_multi')
        return u, None
    check(lib.xw_ode_fwd(_p(xT), _p(t), _p(start), _p(theta), method, N, L, d, H, K, m, _p(u), _p(Y), _stream()), 'xw_ode_fwd')
    return u, Y


def ode_fwd_multi(jobs, t, theta, method, H, K, m, zero16=None, act_x_only=False, narrow=False, prio_drop=0):
    """jobs: list of dicts(xT[d,N], start[N], u[L,N], Y[L,H,N] or None) -- all groups share t, theta; ONE launch.
    act_x_only: the activation stores of this launch will only be read by sweeps without weight gradients (the
    discriminator sub-step): only the tanh rows and the ReLU-mask words are written.
    narrow: narrow tiles (XwOdeFwdJob.narrow, csrc/xw_ode_n4.h): four waves of 4 paths per 16-path tile -- same outputs, same
    store; for launches that leave SIMDs idle."""
    _need_gpu()
    L = t.shape[0]
    d = jobs[0]['xT'].shape[0]
    _chk(t, F64, (L,), 't'); _chk(theta, F64, (theta_size(d, H, K),), 'theta')
    arr = (XwOdeFwdJob * len(jobs))()
    for a, j in zip(arr, jobs):
        N = j['xT'].shape[1]
        _chk(j['xT'], F64, (d, N), 'xT'); _chk(j['start'], F64, (N,), 'start'); _chk(j['u'], F64, (L, N), 'u')
        _chk(j.get('Y'), F64, (L, H, N), 'Y')
        if j.get('act') is not None:
            _chk(j['act'], F64, (max(L - 1, 1), ode_act_rows(method, H, K, m), ode_act_cols(N)), 'act')
        a.xT, a.start, a.u, a.Y, a.act, a.N = _p(j['xT']), _p(j['start']), _p(j['u']), _p(j.get('Y')), _p(j.get('act')), N
        a.act_x_only = 1 if act_x_only else 0
        a.narrow = 1 if narrow else 0
        a.prio_drop = int(prio_drop)
    _chk(zero16, F64, (16,), 'zero16')
    check(lib.xw_ode_fwd_multi(arr, len(jobs), _p(t), _p(theta), method, L, d, H, K, m, _p(zero16), _stream()), 'xw_ode_fwd_multi')


def ode_bwd_multi(jobs, t, theta, method, H, K, m, want_x, want_params, x_cot_ones=False, adjoint=False, narrow=False, prio_drop=0):
    """jobs: list of dicts(xT, start, Y, ubar or None, gx, gs, gslab); ONE launch for all groups.
    res = dict(u[L,N], ref ([N] with first_only, else [L,N]), coef, base, first_only) instead of ubar: the cotangent
    base + coef (u - ref) (at l = 0 only with first_only) is formed inside the sweep.
    x_cot_ones (with want_x and want_params): gx, gs for the all-ones cotangent, parameter gradients for ubar, which must
    equal 1 at every time index >= 1; jobs without gx / gs produce no x outputs.
    adjoint: the continuous adjoint of torchdiffeq.odeint_adjoint (config['adjoint'] = True) instead of the reverse of
    the steps taken; ignores the activation store.
    narrow: narrow tiles (mode bit 4, csrc/xw_ode_n4.h): four waves of 4 paths per 16-path tile instead of one -- for
    launches that leave SIMDs idle; needs every job's activation store (euler, midpoint)."""
    _need_gpu()
    L = t.shape[0]
    d = jobs[0]['xT'].shape[0]
    P = theta_size(d, H, K)
    _chk(t, F64, (L,), 't'); _chk(theta, F64, (P,), 'theta')
    arr = (XwOdeBwdJob * len(jobs))()
    for a, j in zip(arr, jobs):
        N = j['xT'].shape[1]
        _chk(j['xT'], F64, (d, N), 'xT'); _chk(j['start'], F64, (N,), 'start'); _chk(j['Y'], F64, (L, H, N), 'Y')
        _chk(j.get('ubar'), F64, (L, N), 'ubar')
        if want_x and not (x_cot_ones and j.get('gx') is None):
            _chk(j['gx'], F64, (d, N), 'gx'); _chk(j['gs'], F64, (N,), 'gs')
        if want_params:
            _chk(j['gslab'], F64, (ode_bwd_slabs(N), P), 'gslab')
        if j.get('act') is not None:
            _chk(j['act'], F64, (max(L - 1, 1), ode_act_rows(method, H, K, m), ode_act_cols(N)), 'act')
        a.xT, a.start, a.Y, a.ubar, a.N = _p(j['xT']), _p(j['start']), _p(j['Y']), _p(j.get('ubar')), N
        a.act = _p(j.get('act'))
        a.gx, a.gs, a.gslab = _p(j.get('gx')), _p(j.get('gs')), _p(j.get('gslab'))
        res = j.get('res')
        if res is not None:
            # cotangent formed from a residual inside the sweep (XwOdeBwdJob.res_*): base + coef (u - ref)
            if j.get('ubar') is not None:
                raise XnwanError('a sweep job takes a stored cotangent (ubar) or a residual (res), not both')
            if res.get('weak') is not None:
                # the weak form's dI/du: coef d(c(u) u)/du v w (+ base v at the last time index); ref = v
                wk = res['weak']
                wpp = wk['w'].dim() == 2
                _chk(res['u'], F64, (L, N), 'res.u'); _chk(res['ref'], F64, (L, N), 'res.ref (v)')
                _chk(wk['w'], F64, (L, N) if wpp else (N,), 'res.weak.w'); _chk(wk.get('c'), F64, (L, N), 'res.weak.c')
                _chk(wk.get('cp'), F64, (L, N), 'res.weak.cp')
                a.res_first_only, a.res_u, a.res_ref = 2, _p(res['u']), _p(res['ref'])
                a.res_w_per_point, a.res_w, a.res_c, a.res_cp = int(wpp), _p(wk['w']), _p(wk.get('c')), _p(wk.get('cp'))
                a.res_kappa2 = 2.0 * float(wk.get('ckappa', 0.0))
            else:
                first = bool(res['first_only'])
                _chk(res['u'], F64, (L, N), 'res.u'); _chk(res['ref'], F64, (N,) if first else (L, N), 'res.ref')
                a.res_first_only, a.res_u, a.res_ref = int(first), _p(res['u']), _p(res['ref'])
            a.res_coef, a.res_base = float(res['coef']), float(res['base'])
    if x_cot_ones and not (want_x and want_params):
        raise XnwanError('x_cot_ones needs want_x and want_params')
    if x_cot_ones and adjoint:
        raise XnwanError('x_cot_ones is not available with the continuous adjoint')
    if narrow and (adjoint or method > 1 or any(j.get('act') is None for j in jobs)):
        raise XnwanError('narrow-tile sweeps run from the activation store of euler / midpoint (no adjoint=True, no rk4)')
    mode = (1 if want_x else 0) | (2 if want_params else 0) | (4 if x_cot_ones else 0) | (8 if adjoint else 0) | (16 if narrow else 0) | ((int(prio_drop) & 3) << 5)
    check(lib.xw_ode_bwd_multi(arr, len(jobs), _p(t), _p(theta), method, L, d, H, K, m, mode, _stream()), 'xw_ode_bwd_multi')


def ode_bwd_slabs(N):
    return lib.xw_ode_bwd_slabs(N)


def ode_act_cols(N):
    """columns of the activation store for N paths: whole tiles of 16 (the store is tile-major inside a step)"""
    return (N + 15) // 16 * 16


def ode_act_rows(method, H, K, m):
    """rows per step of the activation store that ode_fwd_multi fills and ode_bwd_multi reads (0: not used by `method`)"""
    r = lib.xw_ode_act_rows(int(method), H, K, m)
    check(min(r, 0), 'xw_ode_act_rows')
    return r


def ode_bwd(xT, t, start, theta, Y, ubar, method, H, K, m, want_x=True, want_params=False, gx=None, gs=None, gslab=None,
            adjoint=False):
    """reverse sweep: returns (gx[d,N], gs[N], gslab[nslab,P_u]) -- entries not requested are None."""
    _need_gpu()
    d, N = xT.shape
    L = t.shape[0]
    P = theta_size(d, H, K)
    _chk(xT, F64, (d, N), 'xT'); _chk(t, F64, (L,), 't'); _chk(start, F64, (N,), 'start'); _chk(theta, F64, (P,), 'theta')
    _chk(Y, F64, (L, H, N), 'Y'); _chk(ubar, F64, (L, N), 'ubar')
    mode = (1 if want_x else 0) | (2 if want_params else 0) | (8 if adjoint else 0)
    if want_x:
        gx = torch.empty(d, N, dtype=F64, device=xT.device) if gx is None else gx
        gs = torch.empty(N, dtype=F64, device=xT.device) if gs is None else gs
        _chk(gx, F64, (d, N), 'gx'); _chk(gs, F64, (N,), 'gs')
    if want_params:
        ns = ode_bwd_slabs(N)
        gslab = torch.empty(ns, P, dtype=F64, device=xT.device) if gslab is None else gslab
        _chk(gslab, F64, (ns, P), 'gslab')
    check(lib.xw_ode_bwd(_p(xT), _p(t), _p(start), _p(theta), _p(Y), _p(ubar), method, N, L, d, H, K, m, mode,
                         _p(gx if want_x else None), _p(gs if want_x else None), _p(gslab if want_params else None),
                         _stream()), 'xw_ode_bwd')
    return (gx if want_x else None), (gs if want_x else None), (gslab if want_params else None)


DISC_UNROLLED_DEPTH = 9   # v_layers of the reference's YAML: the depth the recomputing reverse kernels are compiled for


def disc_recompute(W, q):
    """True when the recomputing reverse kernels (xw_disc_gradx, xw_disc_bwd without a record) exist for this container
    width and depth: the reference's YAML shape only.  Everything else reverses from the record / the fused gradient."""
    return W == 50 and q == DISC_UNROLLED_DEPTH


def disc_act_rows(W, q):
    """rows of the activation record disc_fwd can store for disc_bwd: the inputs of the q tied layers + tanh(a_q)"""
    r = lib.xw_disc_act_rows(W, q)
    check(min(r, 0), 'xw_disc_act_rows')
    return r


def disc_act_view(act, W, q):
    """the record as [tiles of 16 points, (q+1) W rows, 16]: row j W + k of a tile = input k of tied layer j (relu(a_j)),
    the last W rows tanh(a_q) -- the layout the kernels use (tile-major; the [rows, cols] shape is only its size)"""
    rows = disc_act_rows(W, q)
    return act.reshape(-1)[:act.numel()].view(act.shape[1] // 16, rows, 16)


def disc_act_cols(P):
    """columns of that record for P points: whole 16-point tiles"""
    return (P + 15) // 16 * 16


def disc_xproj_rows(W):
    """rows of the x-projection table: the row tiles of the width's kernel (64 for the widths 50 and 64, 96 / 128 for the wide containers)"""
    return 128 if W > 96 else 96 if W > 64 else 64


def disc_xproj(xT, phi, W, out=None):
    """the input layer's x-projection per path, [disc_xproj_rows(W), N] (rows >= W zero): Vin[:, 1..d] x_n + Vin.b -- for disc_fwd(xproj=...)."""
    _need_gpu()
    d, N = xT.shape
    _chk(xT, F64, (d, N), 'xT'); _chk(phi, F64, (phi_size(d, W),), 'phi')
    rows = disc_xproj_rows(W)
    out = torch.empty(rows, N, dtype=F64, device=xT.device) if out is None else out
    _chk(out, F64, (rows, N), 'xproj')
    check(lib.xw_disc_xproj(_p(xT), _p(phi), N, d, W, _p(out), _stream()), 'xw_disc_xproj')
    return out


def disc_fwd(xT, t, phi, W, q, tpp=None, want_vt=True, v=None, vt=None, gxv=None, gtv=None, ngrad=0, max_blocks=0, act=None,
             xproj=None):
    """v_phi and dv/dt.  Path mode: points (t[l], x_n) -> [L,N].  Point mode (tpp[N]): points (tpp[n], x_n) -> [1,N].
    gxv[d,ngrad] / gtv[ngrad]: also return the input gradient of v at the leading ngrad points (time-major order).
    act[disc_act_rows(W, q), disc_act_cols(L*N)]: also store the layer inputs, for disc_bwd(act=...).
    xproj[disc_xproj_rows(W),N] (path mode, MFMA widths): disc_xproj's table -- the input layer then costs one load per row and point."""
    _need_gpu()
    d, N = xT.shape
    L = 1 if tpp is not None else t.shape[0]
    _chk(xT, F64, (d, N), 'xT'); _chk(phi, F64, (phi_size(d, W),), 'phi')
    _chk(t, F64, None, 't'); _chk(tpp, F64, (N,), 'tpp')
    v = torch.empty(L, N, dtype=F64, device=xT.device) if v is None else v
    if want_vt and vt is None:
        vt = torch.empty(L, N, dtype=F64, device=xT.device)
    _chk(v, F64, (L, N), 'v'); _chk(vt, F64, (L, N), 'vt')
    if gxv is not None:
        _chk(gxv, F64, (d, ngrad), 'gxv'); _chk(gtv, F64, (ngrad,), 'gtv')
    if act is not None:
        _chk(act, F64, (disc_act_rows(W, q), disc_act_cols(L * N)), 'act')
    if xproj is not None:
        _chk(xproj, F64, (disc_xproj_rows(W), N), 'xproj')
    check(lib.xw_disc_fwd_xproj(_p(xT), _p(t), _p(tpp), _p(phi), N, L, d, W, q, _p(v), _p(vt if want_vt else None), _p(gxv),
                                _p(gtv), int(ngrad), int(max_blocks), _p(act), _p(xproj), _stream()), 'xw_disc_fwd')
    return v, (vt if want_vt else None)


def disc_gradx(xT, t, phi, W, q, tpp=None, vbar=None, gxv=None, gtv=None):
    """input gradient of <vbar, v>: (nabla_x)[d,N] and (d/dt)[N] at the points (tpp[n] or t[0], x_n); vbar None = ones."""
    _need_gpu()
    d, N = xT.shape
    _chk(xT, F64, (d, N), 'xT'); _chk(phi, F64, (phi_size(d, W),), 'phi'); _chk(t, F64, None, 't'); _chk(tpp, F64, (N,), 'tpp')
    gxv = torch.empty(d, N, dtype=F64, device=xT.device) if gxv is None else gxv
    gtv = torch.empty(N, dtype=F64, device=xT.device) if gtv is None else gtv
    _chk(gxv, F64, (d, N), 'gxv'); _chk(gtv, F64, (N,), 'gtv')
    if vbar is not None:
        vbar = vbar.reshape(-1)
        _chk(vbar, F64, (N,), 'vbar')
    if not disc_recompute(W, q):
        # other depths / widths: the forward kernel's fused input gradient (any q <= 16), scaled by the cotangent
        t0 = t[:1] if tpp is None else None
        disc_fwd(xT, t0, phi, W, q, tpp=tpp, want_vt=False, gxv=gxv, gtv=gtv, ngrad=N)
        if vbar is not None:
            gxv.mul_(vbar)
            gtv.mul_(vbar)
        return gxv, gtv
    check(lib.xw_disc_gradx(_p(xT), _p(t), _p(tpp), _p(phi), _p(vbar), N, d, W, q, _p(gxv), _p(gtv), _stream()), 'xw_disc_gradx')
    return gxv, gtv


def disc_bwd_slabs(N, L):
    return lib.xw_disc_bwd_slabs(N, L)


def disc_bwd(xT, t, phi, vbar, W, q, tpp=None, gslab=None, act=None):
    """parameter gradient of <vbar, v> as partial slabs [nslab, P_v].  act: the record disc_fwd stored for the same phi and
    points (skips the forward recompute)."""
    _need_gpu()
    d, N = xT.shape
    L = 1 if tpp is not None else t.shape[0]
    P = phi_size(d, W)
    _chk(xT, F64, (d, N), 'xT'); _chk(phi, F64, (P,), 'phi'); _chk(vbar, F64, (L, N), 'vbar')
    _chk(t, F64, None, 't'); _chk(tpp, F64, (N,), 'tpp')
    ns = disc_bwd_slabs(N, L)
    gslab = torch.empty(ns, P, dtype=F64, device=xT.device) if gslab is None else gslab
    _chk(gslab, F64, (ns, P), 'gslab')
    if act is None and not disc_recompute(W, q):
        # the recomputing kernel exists at the reference's width and depth only: store the record first, then reverse from it
        act = torch.empty(disc_act_rows(W, q), disc_act_cols(L * N), dtype=F64, device=xT.device)
        disc_fwd(xT, t, phi, W, q, tpp=tpp, want_vt=False, act=act)
    if act is not None:
        _chk(act, F64, (disc_act_rows(W, q), disc_act_cols(L * N)), 'act')
    check(lib.xw_disc_bwd(_p(xT), _p(t), _p(tpp), _p(phi), _p(vbar), N, L, d, W, q, _p(act), _p(gslab), _stream()),
          'xw_disc_bwd')
    return gslab


def reduce_work_size():
    return lib.xw_reduce_work_size()


def weak_partials(u, v, vt, w, f, h, Vol, Nglob, scal, work, s3x=None, contract=None, c=None, ckappa=0.0, wt=None,
                  finalize=None, pair=None, bdry=None):
    """partial sums of I, sum v^2, SSE_init.  Either s3x[N] (pre-contracted gradient term) or contract = dict(gx, gs, ghT,
    gxv, w0, gwx0T) for the in-kernel contraction with a = identity, b = 0.
    finalize = dict(Lb, Nbglob, alpha, step[, init_off, bdry_off]) (single GPU): also turn the sums into the loss values
    and advance `step`, exactly what losses() does.
    pair = dict(href[N], s3_scale) (L == 1 only): the reference's [N,N] broadcast on a single-slice T0 group, factorised
    (include/xnwan.h); f must then hold mean(f) in every entry.
    bdry = dict(ub[Lb, Nb], g[Lb, Nb]): also add the boundary sum of squares sum (u_b - g)^2 to scal[3] (bdry_partials' sum, same launch)."""
    _need_gpu()
    L, N = u.shape
    for name, a in (('u', u), ('v', v), ('vt', vt), ('f', f)):
        _chk(a, F64, (L, N), name)
    per_point = 1 if w.dim() == 2 else 0
    _chk(w, F64, (L, N) if per_point else (N,), 'w'); _chk(wt, F64, (L, N), 'wt'); _chk(c, F64, (L, N), 'c')
    _chk(s3x, F64, (N,), 's3x'); _chk(h, F64, (N,), 'h'); _chk(scal, F64, (16,), 'scal')
    _chk(work, F64, (reduce_work_size(),), 'work')
    fz = finalize or {}
    _chk(fz.get('step'), torch.int64, (1,), 'step')
    bd, Pb = bdry or {}, 0
    if bdry is not None:
        Pb = bdry['ub'].numel()
        _chk(bdry['ub'], F64, tuple(bdry['ub'].shape), 'ub'); _chk(bdry['g'], F64, tuple(bdry['ub'].shape), 'g')
    pr = pair or {}
    _chk(pr.get('href'), F64, (N,), 'href')
    if pair is not None and L != 1:
        raise XnwanError('the pairwise form only exists for single-slice groups (L == 1)')
    k, d = {}, 0
    if s3x is None:
        k = contract
        d = k['gx'].shape[0]
        for name in ('gx', 'ghT', 'gxv', 'gwx0T'):
            _chk(k[name], F64, (d, N), name)
        _chk(k['gs'], F64, (N,), 'gs'); _chk(k['w0'], F64, (N,), 'w0')
    check(lib.xw_weak_partials(_p(u), _p(v), _p(vt), _p(w), per_point, _p(wt), _p(s3x), _p(k.get('gx')), _p(k.get('gs')),
                               _p(k.get('ghT')), _p(k.get('gxv')), _p(k.get('w0')), _p(k.get('gwx0T')), d, _p(c),
                               float(ckappa), _p(f), _p(h), _p(pr.get('href')), 0 if pair is None else 1,
                               float(pr.get('s3_scale', 1.0)), N, L, float(Vol), float(Nglob), _p(work), _p(scal),
                               0 if finalize is None else 1, max(int(fz.get('Lb', 1)), 1), float(fz.get('Nbglob', 1.0)),
                               float(fz.get('alpha', 0.0)), float(fz.get('init_off', 0.0)), float(fz.get('bdry_off', 0.0)),
                               _p(fz.get('step')), _p(bd.get('ub')), _p(bd.get('g')), Pb, _stream()),
          'xw_weak_partials')


def pair_fold(scal, Vol, Nglob):
    """scal[0] -= (Vol / Nglob) scal[7] scal[8] (several GPUs: after the all-reduce of the partial sums of a pairwise group)"""
    _need_gpu()
    _chk(scal, F64, (16,), 'scal')
    check(lib.xw_pair_fold(_p(scal), float(Vol), float(Nglob), _stream()), 'xw_pair_fold')


def cube_weight(x, top, bot, w, gwT, w0=None, xT=None):
    """w[N], dw/dx as gwT[d, N] (float64) of the hypercube's distance weight at the float32 points x[N, d] -- the values and the
    gradient Hypercube.func_w / autograd give in float32, widened; optionally w0 (a second copy of w) and xT = x^T as float64"""
    _need_gpu()
    N, d = x.shape
    _chk(x, torch.float32, (N, d), 'x')
    _chk(w, F64, (N,), 'w')
    _chk(gwT, F64, (d, N), 'gwT')
    if w0 is not None:
        _chk(w0, F64, (N,), 'w0')
    if xT is not None:
        _chk(xT, F64, (d, N), 'xT')
    check(lib.xw_cube_weight(_p(x), N, d, float(top), float(bot), _p(w), _p(w0), _p(gwT), _p(xT), _stream()), 'xw_cube_weight')


GATHER_ROWS = 1024        # rows of one xw_gather_fields table (include/xnwan.h)


def gather_fields(table, count, total):
    """dst[i][j][k] = src[i s0 + j s1 + k s2] for every row (src, dst, n0, n1, n2, s0, s1, s2, before) of the int64 device table
    [>= count, 9] (addresses as integers; float64 arrays; dst contiguous): the sample fields of all groups of a list-domain
    sample in one launch (Engine.load_groups_packed)"""
    _need_gpu()
    if table.dtype != torch.int64 or table.dim() != 2 or table.shape[1] != 9 or not table.is_contiguous() or table.shape[0] < count:
        raise XnwanError('gather table: need a contiguous int64 [>= count, 9] device tensor')
    check(lib.xw_gather_fields(_p(table), int(count), int(total), _stream()), 'xw_gather_fields')


def weak_contract_general(A0, amode, B0, gx, gs, ghT, gxv, w0, gwx0T, v0, s3x):
    """s3x[n] = sum_ij a_ij d_i(phi) d_j(u) + phi sum_i b_i d_i(u) at the first time index for general coefficients.
    amode says what A0 is (never inferred from its shape: [d,d] and [d,N] coincide when N == d):
    0 identity (A0 None), 1 [d,d] one matrix, 2 [d,N] diagonal, 3 [d,d,N] full table at t_0;  B0: None or [d,N]."""
    _need_gpu()
    d, N = gx.shape
    for name, a in (('gx', gx), ('ghT', ghT), ('gxv', gxv), ('gwx0T', gwx0T)):
        _chk(a, F64, (d, N), name)
    for name, a in (('gs', gs), ('w0', w0), ('v0', v0), ('s3x', s3x)):
        _chk(a, F64, (N,), name)
    _chk(B0, F64, (d, N), 'B0')
    amode = int(amode)
    if amode not in (0, 1, 2, 3) or (amode == 0) != (A0 is None):
        raise XnwanError('weak_contract_general: amode %r does not match A0' % (amode,))
    if amode:
        _chk(A0, F64, {1: (d, d), 2: (d, N), 3: (d, d, N)}[amode], 'A0')
    check(lib.xw_weak_contract_general(_p(A0), amode, _p(B0), _p(gx), _p(gs), _p(ghT), _p(gxv), _p(w0), _p(gwx0T), _p(v0), d, N,
                                       _p(s3x), _stream()), 'xw_weak_contract_general')
    return s3x


def bdry_partials(ub, g, alpha, Nbglob, scal, work, ubar_b=None):
    _need_gpu()
    L, Nb = ub.shape
    _chk(ub, F64, (L, Nb), 'ub'); _chk(g, F64, (L, Nb), 'g'); _chk(ubar_b, F64, (L, Nb), 'ubar_b'); _chk(scal, F64, (16,), 'scal')
    _chk(work, F64, (reduce_work_size(),), 'work')
    check(lib.xw_bdry_partials(_p(ub), _p(g), Nb, L, float(alpha), float(Nbglob), _p(ubar_b), _p(work), _p(scal), _stream()),
          'xw_bdry_partials')


def gen_cotangents(u, v, w, h, Vol, Nglob, alpha, ubarA, ubarB, c=None, cp=None, ckappa=0.0, pollution=1.0, scal=None):
    """cotangent bases of the generator loss; pass ubarA=None or ubarB=None to form only one of them"""
    _need_gpu()
    if ubarA is None and ubarB is None:
        raise XnwanError('gen_cotangents: nothing to compute')
    L, N = u.shape
    per_point = 1 if w.dim() == 2 else 0
    _chk(u, F64, (L, N), 'u'); _chk(v, F64, (L, N), 'v'); _chk(w, F64, (L, N) if per_point else (N,), 'w')
    _chk(c, F64, (L, N), 'c'); _chk(cp, F64, (L, N), 'cp'); _chk(h, F64, (N,), 'h')
    _chk(ubarA, F64, (L, N), 'ubarA'); _chk(ubarB, F64, (L, N), 'ubarB'); _chk(scal, F64, (16,), 'scal')
    check(lib.xw_gen_cotangents(_p(u), _p(v), _p(w), per_point, _p(c), _p(cp), float(ckappa), _p(h), N, L, float(Vol),
                                float(Nglob), float(alpha), float(pollution), _p(scal), _p(ubarA), _p(ubarB), _stream()),
          'xw_gen_cotangents')


def disc_cotangent(u, v, w, f, h, Vol, Nglob, scal, vbar, c=None, ckappa=0.0, pollution=1.0, s3_scale=1.0):
    _need_gpu()
    L, N = u.shape
    per_point = 1 if w.dim() == 2 else 0
    _chk(u, F64, (L, N), 'u'); _chk(v, F64, (L, N), 'v'); _chk(w, F64, (L, N) if per_point else (N,), 'w')
    _chk(c, F64, (L, N), 'c'); _chk(f, F64, (L, N), 'f'); _chk(h, F64, (N,), 'h'); _chk(vbar, F64, (L, N), 'vbar')
    _chk(scal, F64, (16,), 'scal')
    check(lib.xw_disc_cotangent(_p(u), _p(v), _p(w), per_point, _p(c), float(ckappa), _p(f), _p(h), N, L, float(Vol),
                                float(Nglob), float(pollution), float(s3_scale), _p(scal), _p(vbar), _stream()), 'xw_disc_cotangent')


def losses(scal, L, Lb, Vol, Nglob, Nbglob, alpha, step=None, init_off=0.0, bdry_off=0.0):
    """loss values from the partial sums; also increments `step` when given (pair with adam(..., bump_step=False))"""
    _need_gpu()
    _chk(scal, F64, (16,), 'scal'); _chk(step, torch.int64, (1,), 'step')
    check(lib.xw_losses(_p(scal), L, max(int(Lb), 1), float(Vol), float(Nglob), float(Nbglob), float(alpha), float(init_off),
                        float(bdry_off), _p(step), _stream()), 'xw_losses')


def adam(param, gslabA, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, gextraA=None, gslabB=None, gextraB=None,
         scal=None, gsum_out=None, bump_step=True, lag=None, lag_range=(0, 0), skip=False):
    """param -= Adam(g),  g = gextraA + sum(gslabA) + coefB (gextraB + sum(gslabB)),  coefB = 2 / scal[0] if scal else 1
    lag (device int64[1]) / lag_range: parameters [lo, hi) count their own steps, step - lag; skip: leave that range
    untouched this time and advance lag (torch's Adam skipping parameters whose .grad is None)"""
    _need_gpu()
    P = param.shape[0]
    _chk(param, F64, (P,), 'param'); _chk(m, F64, (P,), 'm'); _chk(v, F64, (P,), 'v'); _chk(step, torch.int64, (1,), 'step')
    nA = nB = 0
    if gslabA is not None:
        nA = gslabA.shape[0]
        _chk(gslabA, F64, (nA, P), 'gslabA')
    if gslabB is not None:
        nB = gslabB.shape[0]
        _chk(gslabB, F64, (nB, P), 'gslabB')
    _chk(gextraA, F64, (P,), 'gextraA'); _chk(gextraB, F64, (P,), 'gextraB'); _chk(gsum_out, F64, (P,), 'gsum_out')
    _chk(scal, F64, (16,), 'scal'); _chk(lag, torch.int64, (1,), 'lag')
    lo, hi = (int(lag_range[0]), int(lag_range[1])) if lag is not None else (0, 0)
    check(lib.xw_adam(_p(param), _p(gslabA), nA, _p(gextraA), _p(gslabB), nB, _p(gextraB), _p(scal), _p(m), _p(v), _p(step),
                      (1 if bump_step is True else 0 if bump_step is False else int(bump_step)), P,
                      float(lr), float(beta1), float(beta2), float(eps), _p(gsum_out), lo, hi, 1 if skip else 0, _p(lag),
                      _stream()), 'xw_adam')


def slab_sum2(gA, outA, gB, outB):
    """outA = sum of the slabs gA[nA, P], outB = sum of gB[nB, P], one launch"""
    _need_gpu()
    (nA, P), (nB, P2) = gA.shape, gB.shape
    _chk(gA, F64, (nA, P), 'gA'); _chk(gB, F64, (nB, P), 'gB'); _chk(outA, F64, (P,), 'outA'); _chk(outB, F64, (P,), 'outB')
    check(lib.xw_slab_sum2(_p(gA), nA, _p(outA), _p(gB), nB, _p(outB), P, _stream()), 'xw_slab_sum2')


def slab_sum(gslab, out=None, accumulate=False):
    _need_gpu()
    ns, P = gslab.shape
    _chk(gslab, F64, (ns, P), 'gslab')
    out = torch.empty(P, dtype=F64, device=gslab.device) if out is None else out
    _chk(out, F64, (P,), 'out')
    check(lib.xw_slab_sum(_p(gslab), ns, P, 1 if accumulate else 0, _p(out), _stream()), 'xw_slab_sum')
    return out
