"""ctypes binding of libxnwan.so (C ABI declared in include/xnwan.h).

The library is the product: there is no CPU or eager-PyTorch fallback.  Importing this module without the built
library raises, and so does any call when no GPU is present.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('XW_LIBRARY') or os.path.join(_HERE, 'libxnwan.so')   # (override: kernel experiments only)
ABI_VERSION = 31

c_f32p = ctypes.c_void_p   # coordinates / time grid: const double* (device)   [name kept from the float32 era]
c_f64p = ctypes.c_void_p   # double*       (device)
c_i64p = ctypes.c_void_p
c_int, c_dbl, c_vp = ctypes.c_int, ctypes.c_double, ctypes.c_void_p


class XwOdeFwdJob(ctypes.Structure):      # include/xnwan.h
    _fields_ = [('xT', c_vp), ('start', c_vp), ('u', c_vp), ('Y', c_vp), ('act', c_vp), ('N', c_int), ('act_x_only', c_int),
                ('narrow', c_int), ('prio_drop', c_int)]


class XwOdeBwdJob(ctypes.Structure):
    _fields_ = [('xT', c_vp), ('start', c_vp), ('Y', c_vp), ('act', c_vp), ('ubar', c_vp), ('gx', c_vp), ('gs', c_vp),
                ('gslab', c_vp), ('N', c_int), ('res_first_only', c_int), ('res_u', c_vp), ('res_ref', c_vp),
                ('res_coef', ctypes.c_double), ('res_base', ctypes.c_double), ('res_w_per_point', c_int), ('res_w', c_vp),
                ('res_c', c_vp), ('res_cp', c_vp), ('res_kappa2', ctypes.c_double)]


class XwGroup(ctypes.Structure):           # include/xnwan.h: one group of paths for xw_substep_gen / xw_substep_disc
    _fields_ = ([(n, c_int) for n in ('N', 'Nb', 'L', 'Lb', 'd', 'same_grid', 'w_per_point', 'amode', 'pair_i', 'pair_b', 'ns_u', 'ns_b',
                                      'narrow', 'sharded')]
                + [(n, c_dbl) for n in ('Vol', 'Nglob', 'Nbglob', 's3_scale', 'init_off', 'bdry_off', 'ckappa')]
                + [(n, c_vp) for n in ('xT', 'xvT', 'xbT', 't', 'tb', 'tpp', 'xvT_pts', 'start', 'start_b', 'h', 'href', 'f', 'g', 'w',
                                       'wt', 'w0', 'ghT', 'gwx0T', 'c', 'cp', 'A0', 'B0', 'u', 'ub', 'Y', 'Yb', 'act', 'act_b', 'v',
                                       'vt', 'gxv', 'gtv', 'gx', 'gs', 'vbar', 's3x', 'vact', 'slabA', 'slabB', 'slab_v', 'work_i',
                                       'work_b', 'xproj')])


class XwSolverState(ctypes.Structure):
    _fields_ = ([(n, c_int) for n in ('method', 'H', 'K', 'm', 'W', 'q', 'Pu', 'Pv', 'adjoint', 'v_blocks', 'v_blocks_disc', 'lag_lo',
                                      'lag_hi')]
                + [(n, c_dbl) for n in ('alpha', 'pollution', 'lr_u', 'lr_v', 'beta1', 'beta2', 'eps')]
                + [(n, c_vp) for n in ('theta', 'phi', 'scal', 'grad_u', 'grad_v', 'm_u', 'v_u', 'm_v', 'v_v', 'step_u', 'step_v',
                                       'lag_u', 'exchange', 'exchange_ctx', 'pack_u')])


# XwSolverState.exchange: int (*)(double* buf, int count, void* ctx, void* stream) -- xw_allreduce's own signature
EXCHANGE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p)


# name -> argument types (return type is always int); mirrors include/xnwan.h line by line
SIGNATURES = {
    'xw_abi_version': [],
    'xw_supported_dims': [ctypes.c_char_p, c_int],
    'xw_theta_size': [c_int, c_int, c_int],
    'xw_phi_size': [c_int, c_int],
    'xw_ode_fwd': [c_f32p, c_f32p, c_f64p, c_f64p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_f64p, c_f64p, c_vp],
    'xw_ode_fwd_multi': [ctypes.POINTER(XwOdeFwdJob), c_int, c_f32p, c_f64p, c_int, c_int, c_int, c_int, c_int, c_int, c_f64p,
                         c_vp],
    'xw_ode_bwd_slabs': [c_int],
    'xw_ode_act_rows': [c_int, c_int, c_int, c_int],
    'xw_ode_bwd_multi': [ctypes.POINTER(XwOdeBwdJob), c_int, c_f32p, c_f64p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp],
    'xw_ode_bwd': [c_f32p, c_f32p, c_f64p, c_f64p, c_f64p, c_f64p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                   c_f64p, c_f64p, c_f64p, c_vp],
    'xw_disc_fwd': [c_f32p, c_f32p, c_f32p, c_f64p, c_int, c_int, c_int, c_int, c_int, c_f64p, c_f64p, c_f64p, c_f64p, c_int,
                    c_int, c_f64p, c_vp],
    'xw_disc_xproj': [c_f32p, c_f64p, c_int, c_int, c_int, c_f64p, c_vp],
    'xw_disc_fwd_xproj': [c_f32p, c_f32p, c_f32p, c_f64p, c_int, c_int, c_int, c_int, c_int, c_f64p, c_f64p, c_f64p, c_f64p, c_int,
                          c_int, c_f64p, c_f64p, c_vp],
    'xw_disc_act_rows': [c_int, c_int],
    'xw_disc_gradx': [c_f32p, c_f32p, c_f32p, c_f64p, c_f64p, c_int, c_int, c_int, c_int, c_f64p, c_f64p, c_vp],
    'xw_disc_bwd_slabs': [c_int, c_int],
    'xw_disc_bwd': [c_f32p, c_f32p, c_f32p, c_f64p, c_f64p, c_int, c_int, c_int, c_int, c_int, c_f64p, c_f64p, c_vp],
    'xw_mt19937_uniform_f32': [c_vp, ctypes.c_long, c_vp, ctypes.c_long, ctypes.c_float, ctypes.c_float, c_int],
    'xw_gather_fields': [c_vp, c_int, ctypes.c_long, c_vp],
    'xw_mt19937_legacy_normal_f64': [c_vp, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_dbl), c_vp, ctypes.c_long],
    'xw_weak_partials': [c_f64p, c_f64p, c_f64p, c_f64p, c_int, c_f64p, c_f64p, c_f64p, c_f64p, c_f64p, c_f64p, c_f64p,
                         c_f64p, c_int, c_f64p, c_dbl, c_f64p, c_f64p, c_f64p, c_int, c_dbl, c_int, c_int, c_dbl, c_dbl, c_f64p, c_f64p,
                         c_int, c_int, c_dbl, c_dbl, c_dbl, c_dbl, c_i64p, c_f64p, c_f64p, ctypes.c_long, c_vp],
    'xw_pair_fold': [c_f64p, c_dbl, c_dbl, c_vp],
    'xw_cube_weight': [c_vp, c_int, c_int, c_dbl, c_dbl, c_f64p, c_f64p, c_f64p, c_f64p, c_vp],
    'xw_bdry_partials': [c_f64p, c_f64p, c_int, c_int, c_dbl, c_dbl, c_f64p, c_f64p, c_f64p, c_vp],
    'xw_reduce_work_size': [],
    'xw_weak_contract_general': [c_f64p, c_int, c_f64p, c_f64p, c_f64p, c_f64p, c_f64p, c_f64p, c_f64p, c_f64p, c_int, c_int, c_f64p, c_vp],
    'xw_gen_cotangents': [c_f64p, c_f64p, c_f64p, c_int, c_f64p, c_f64p, c_dbl, c_f64p, c_int, c_int, c_dbl, c_dbl, c_dbl,
                          c_dbl, c_f64p, c_f64p, c_f64p, c_vp],
    'xw_disc_cotangent': [c_f64p, c_f64p, c_f64p, c_int, c_f64p, c_dbl, c_f64p, c_f64p, c_int, c_int, c_dbl, c_dbl, c_dbl,
                          c_dbl, c_f64p, c_f64p, c_vp],
    'xw_losses': [c_f64p, c_int, c_int, c_dbl, c_dbl, c_dbl, c_dbl, c_dbl, c_dbl, c_i64p, c_vp],
    'xw_adam': [c_f64p, c_f64p, c_int, c_f64p, c_f64p, c_int, c_f64p, c_f64p, c_f64p, c_f64p, c_i64p, c_int, c_int, c_dbl,
                c_dbl, c_dbl, c_dbl, c_f64p, c_int, c_int, c_int, c_i64p, c_vp],
    'xw_substep_gen': [ctypes.POINTER(XwGroup), ctypes.POINTER(XwSolverState), c_int, c_int, c_f64p, c_int, c_vp],
    'xw_substep_disc': [ctypes.POINTER(XwGroup), ctypes.POINTER(XwSolverState), c_int, c_int, c_f64p, c_vp],
    'xw_slab_sum': [c_f64p, c_int, c_int, c_int, c_f64p, c_vp],
    'xw_slab_sum2': [c_f64p, c_int, c_f64p, c_f64p, c_int, c_f64p, c_int, c_vp],
    'xw_comm_available': [],
    'xw_comm_unique_id': [ctypes.c_char_p],
    'xw_comm_init': [ctypes.c_char_p, c_int, c_int, ctypes.POINTER(c_vp)],
    'xw_allreduce': [c_f64p, c_int, c_vp, c_vp],
    'xw_comm_destroy': [c_vp],
}

ERRORS = {-1: 'XW_E_DIMS: network widths/depths not among the compiled kernel instantiations',
          -2: 'XW_E_ARG: null pointer, non-positive size or bad enum',
          -3: 'XW_E_WORKSPACE: workspace too small',
          -4: 'XW_E_COMM: RCCL is not available in this process, or a collective call failed'}


class XnwanError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise XnwanError(
            'libxnwan.so not found at %s -- build it with `python -c "import __graft_entry__ as g; g.build()"` '
            'or `make -C xnode_wan_pde_solver_amd/csrc`.  There is no CPU fallback.' % LIB_PATH)
    # torch FIRST: its wheel ships its own ROCm runtime (torch/lib/libamdhip64.so); loaded before torch, this library would pull in
    # the system's copy and the process would hold two HIP runtimes -- its kernels registered with one, torch's streams and memory
    # living in the other (every launch then fails with hipErrorNoDevice: build() followed by smoke() in ONE process did)
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)           # AttributeError here = the .so does not export the declared ABI
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    got = lib.xw_abi_version()
    if got != ABI_VERSION:
        raise XnwanError('libxnwan.so ABI version %d, host expects %d -- rebuild' % (got, ABI_VERSION))
    return lib


lib = _load()


def check(status, what):
    if status == 0:
        return
    if status in ERRORS:
        buf = ctypes.create_string_buffer(512)
        lib.xw_supported_dims(buf, 512)
        raise XnwanError('%s: %s (compiled: %s)' % (what, ERRORS[status], buf.value.decode()))
    raise XnwanError('%s: HIP launch error %d' % (what, status))
