"""CPU-side checks of the product's host logic (no kernel is launched): parameter parsing, the RNG draw order of
network construction and sampling against the reference-generated golden vectors, state_dict key layout, coefficient
structure probing, and that the built C-ABI library exports exactly what include/xnwan.h declares."""
import json
import os
import re
import subprocess

import numpy as np
import pytest
import torch

import configs.Ex4_1_funcs as P
from xnode_wan_pde_solver_amd import sampling, solver as S, _lib
from xnode_wan_pde_solver_amd.engine import Structure

CASES = ['ref_tiny_midpoint', 'ref_plumb_midpoint', 'ref_d20_small_midpoint', 'ref_d50_nt64_small_midpoint',
         'ref_d100_small_midpoint']
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(golden_dir, case):
    z = np.load(os.path.join(golden_dir, case + '.npz'))
    return z, json.loads(str(z['params_json']))


def test_params_are_read_by_key_and_notebook_dict_is_accepted():
    z_params = {'domain': 'Hypercube', 'iterations': 7, 'solver': 'midpoint', 'adjoint': False, 'min_steps': 5,
                'v_rate': 0.04, 'u_rate': 0.015, 'n2': 1, 'n1': 2, 'v_hidden_dim': 50, 'v_layers': 9,
                'u_hidden_hidden_dim': 10, 'u_hidden_dim': 20, 'u_layers': 8, 'alpha': 1e4,
                'dim': 5, 'N_t': 20, 'N_r': 400, 'N_b': 400, 'T0': 0, 'T': 1}          # shuffled order, no shape_param
    config, setup, iters = S.split_params(z_params)
    assert list(config) == S.CONFIG_KEYS and iters == 7
    assert setup['shape_param'] == [-1, 1] and setup['N_r'] == 400
    with pytest.raises(KeyError):
        S.split_params({k: v for k, v in z_params.items() if k != 'u_rate'})
    assert sampling.resolve_domain('Hypercube') is sampling.Hypercube
    with pytest.raises(KeyError):
        sampling.resolve_domain('__import__("os")')


@pytest.mark.parametrize('case', CASES)
def test_network_construction_draws_like_the_reference(golden_dir, case):
    z, params = load(golden_dir, case)
    config, setup, _ = S.split_params(params)
    torch.manual_seed(int(z['seed']))
    u_net, v_net = S.build_networks(config, setup, P.func_h, P.func_g, sampling.Hypercube)
    for tag, net in (('u', u_net), ('v', v_net)):
        sd = net.state_dict()
        assert list(sd.keys()) == [str(k) for k in z[tag + '_sd_keys']]
        named = dict(net.named_parameters())
        assert list(named) == [str(k) for k in z[tag + '_param_names']]
        for key, canon in zip(z[tag + '_sd_keys'], z[tag + '_sd_alias_of']):
            assert sd[str(key)].data_ptr() == named[str(canon)].data_ptr()        # tied-layer aliases
        for n, p in named.items():
            assert p.dtype == torch.float64
            assert np.array_equal(p.detach().numpy(), z[tag + '_sd/' + n]), n
    # the sample of the first outer iteration continues the same RNG stream
    domain = sampling.Hypercube(setup['shape_param'], setup['dim'], setup['T0'], setup['T'], setup['N_t'])
    pts = sampling.Comb_loader(setup['N_r'], setup['N_b'], domain, torch.device('cpu'))
    assert np.array_equal(domain.times.numpy(), z['times'])
    assert np.array_equal(pts.interioru[:, 0, 1:].detach().numpy(), z['x_u'])
    assert np.array_equal(pts.interiorv[:, 0, 1:].detach().numpy(), z['x_v'])
    assert np.array_equal(pts.boundary[:, 0, 1:].detach().numpy(), z['x_b'])
    if 'X' in z.files:
        assert np.array_equal(pts.interioru.detach().numpy(), z['X'])
        assert np.array_equal(pts.boundary.detach().numpy(), z['BX'])
    assert np.array_equal(domain.func_w(pts.interiorv).detach().numpy(), z['w_v'])
    assert domain.V() == float(z['V'])
    groups = list(pts)
    assert len(groups) == 1 and len(pts) == 1 and all(t.requires_grad for t in groups[0])


def test_boundary_points_sit_on_faces():
    torch.manual_seed(3)
    dom = sampling.Hypercube([-1, 1], 4, 0, 1, 5)
    bx = dom.boundary(37)[:, 0, 1:]
    assert bx.shape == (37, 4)
    assert bool(torch.all((bx.abs() == 1).sum(1) >= 1))
    assert float(dom.func_w(dom.boundary(16)).abs().max()) == 0.0
    assert dom.interior(9).shape == (9, 5, 5)
    with pytest.raises(AssertionError):
        sampling.Hypercube([1, 1], 2, 0, 1, 4)


def test_fillt_matches_reference_vectors(golden_dir):
    z = np.load(os.path.join(golden_dir, 'ref_fillt.npz'))
    for k in range(int(z['n'])):
        idx, filled = sampling.fillt(torch.tensor(z['%d/t' % k]), 1.0, 0.0, int(z['%d/ms' % k]))
        assert np.array_equal(idx.numpy(), z['%d/idx' % k]) and np.array_equal(filled.numpy(), z['%d/filled' % k]), k


def test_structure_probe():
    f = dict(a=P.func_a, b=P.func_b, c=P.func_c)
    st = Structure(f, 6)
    assert st.a_identity and st.b_zero and st.c_kappa == -1.0
    f2 = dict(a=lambda X, i, j: (1.0 + X[..., 1] ** 2) * (i == j), b=lambda X, i: X[..., 0] * (i == 0),
              c=lambda X, u: u ** 3)
    st2 = Structure(f2, 3)
    assert not st2.a_identity and not st2.b_zero and st2.c_kappa is None


def test_cabi_exports_match_header():
    hdr = open(os.path.join(ROOT, 'include', 'xnwan.h')).read()
    declared = set(re.findall(r'^\s*int\s+(xw_\w+)\s*\(', hdr, flags=re.M))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    out = subprocess.run(['nm', '-D', '--defined-only', _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r'\bT\s+(xw_\w+)', out))
    assert declared <= exported, declared - exported
    # ... and nothing else: no kernel launch stub or helper leaks into the dynamic symbol table
    assert set(re.findall(r'\bT\s+(\S+)', out)) <= exported | {'_init', '_fini'}, set(re.findall(r'\bT\s+(\S+)', out)) - exported
    # argument counts of the ctypes signatures agree with the header
    for name, args in _lib.SIGNATURES.items():
        m = re.search(r'int\s+' + name + r'\s*\((.*?)\)\s*;', hdr, flags=re.S)
        n_hdr = 0 if m.group(1).strip() == 'void' else len(m.group(1).split(','))
        assert n_hdr == len(args), name
    # the job structs: same field names in the same order as the ctypes mirrors
    for cname, ctype in (('XwOdeFwdJob', _lib.XwOdeFwdJob), ('XwOdeBwdJob', _lib.XwOdeBwdJob)):
        body = re.search(r'typedef struct \{([^}]*)\}\s*' + cname + r'\s*;', hdr).group(1)
        body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
        fields = [re.split(r'[\s\*]+', decl.strip())[-1] for decl in body.split(';') if decl.strip()]
        assert fields == [f[0] for f in ctype._fields_], (cname, fields)
    # the sub-step runner's structs (several declarators per statement): every declared name, in order
    for cname, ctype in (('XwGroup', _lib.XwGroup), ('XwSolverState', _lib.XwSolverState)):
        body = re.search(r'typedef struct \{([^}]*)\}\s*' + cname + r'\s*;', hdr).group(1)
        body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
        fields = [re.sub(r'[\s\*]', '', name) for decl in body.split(';') if decl.strip()
                  for name in re.sub(r'^\s*(const\s+)?(long\s+long|double|int|void|XwExchangeFn)\b', '', decl.strip()).split(',')]
        assert fields == [f[0] for f in ctype._fields_], (cname, fields)
    # host-side (no GPU) entry points are callable
    assert _lib.lib.xw_abi_version() == _lib.ABI_VERSION
    assert _lib.lib.xw_theta_size(20, 20, 10) == 1651 and _lib.lib.xw_phi_size(20, 50) == 3701


def test_no_gpu_means_loud_failure():
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from xnode_wan_pde_solver_amd import kernels as KN
    with pytest.raises(_lib.XnwanError):
        KN.ode_fwd(torch.zeros(3, 16), torch.zeros(4), torch.zeros(16, dtype=torch.float64),
                   torch.zeros(KN.theta_size(3, 20, 10), dtype=torch.float64), 1, 20, 10, 8)
    params = json.loads(str(np.load(os.path.join(ROOT, 'tests', 'golden', 'ref_tiny_midpoint.npz'))['params_json']))
    with pytest.raises(_lib.XnwanError):
        S.NODE_WAN_solver(params, P.func_a, P.func_b, P.func_c, P.func_h, P.func_f, P.func_g, torch.device('cpu'), './')


def test_product_never_imports_the_oracle():
    bad = []
    for base in ('xnode_wan_pde_solver_amd', 'src', 'utils', 'NODE_WAN_model', 'configs'):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for fn in files:
                if fn.endswith(('.py', '.hip', '.h', '.cpp')):
                    txt = open(os.path.join(dirpath, fn)).read()
                    if re.search(r'^\s*(from|import)\s+oracle\b', txt, flags=re.M) or 'refspec' in txt:
                        bad.append(os.path.join(dirpath, fn))
    assert not bad, bad
    main_txt = open(os.path.join(ROOT, 'main.py')).read()
    assert 'oracle' not in main_txt


@pytest.mark.parametrize('case,name', [('ref_cone_groups', 'NSphere_TCone'), ('ref_hourglass_groups', 'NSphere_THourglass'),
                                       ('ref_cone_ex43_d10_groups', 'NSphere_TCone'),              # BASELINE configs[4]:
                                       ('ref_hourglass_ex43_d10_groups', 'NSphere_THourglass'),    # d = 10, Ex4_3 functions
                                       # a radius that is not a power of two: r * (float32 expr).double() != (r * float32 expr).double()
                                       ('ref_hourglass_r07_sampling', 'NSphere_THourglass'), ('ref_cone_r07_sampling', 'NSphere_TCone')])
def test_sphere_domains_sample_like_the_reference(golden_dir, case, name):
    z, params = load(golden_dir, case)
    params.pop('funcs', None)
    config, setup, _ = S.split_params(params)
    torch.manual_seed(int(z['seed']))
    np.random.seed(int(z['seed']))
    dom_cls = sampling.resolve_domain(name)
    S.build_networks(config, setup, P.func_h, P.func_g, dom_cls)          # consumes the RNG like the constructor
    domain = dom_cls(setup['shape_param'], setup['dim'], setup['T0'], setup['T'], setup['N_t'])
    pts = sampling.Comb_loader(setup['N_r'], setup['N_b'], domain, torch.device('cpu'))
    assert np.array_equal(domain.times.numpy(), z['times']) and domain.V() == float(z['V'])
    assert len(pts.interioru) == int(z['n_interior']) and len(pts.boundary) == int(z['n_boundary'])
    for k, g in enumerate(pts.interioru):
        assert g.dtype == torch.float64 and g.requires_grad
        assert np.array_equal(g.detach().numpy(), z['interior/%d' % k])
        assert np.array_equal(pts.interiorv[k].detach().numpy(), z['interior/%d' % k])       # v sample = copy of the u sample
        assert np.array_equal(domain.func_w(g).detach().numpy(), z['w/%d' % k])
    for k, g in enumerate(pts.boundary):
        assert np.array_equal(g.detach().numpy(), z['boundary/%d' % k])
        assert float(domain.func_w(g.detach()).abs().max()) < 1e-12                          # boundary points sit on the boundary
    assert len(list(pts)) == min(len(pts.interioru), len(pts.boundary))                      # silent truncation (Q7)


def test_incremental_json_files_are_byte_identical_to_json_dump(tmp_path):
    """losses_NODE_{d}.json / Time_NODE_{d}.json are rewritten after every sub-iteration (src/training.py:133-134,166-167);
    the host loop formats them incrementally, the bytes on disk must be those of json.dump(whole list)"""
    import json
    from xnode_wan_pde_solver_amd.solver import _JsonList
    g = torch.Generator().manual_seed(3)
    vals = [1.5, 1e-300, float('inf'), float('nan'), -0.0, 7, 1e22, 0.1 + 0.2]
    vals += [float(x) * 10.0 ** int(e) for x, e in zip(torch.rand(64, generator=g, dtype=torch.float64),
                                                        torch.randint(-6, 9, (64,), generator=g))]
    got, ref = _JsonList([0.25]), [0.25]
    path = tmp_path / 'l.json'
    for x in vals:
        got.append(x); ref.append(x)
        got.write(path)
        assert path.read_text() == json.dumps(ref)
    assert list(got) == ref or all(a == b or (a != a and b != b) for a, b in zip(got, ref))
    empty = _JsonList()
    empty.write(path)
    assert path.read_text() == '[]'


def test_custom_operators_are_registered():
    """north_star: 'driven from Python through PyTorch-ROCm custom ops' -- the module-level call surface dispatches to
    torch.library operators (ops.py); here only their registration and schemas (no GPU)"""
    from xnode_wan_pde_solver_amd import ops
    for name in ('xnode_forward', 'xnode_backward', 'testnet_forward', 'testnet_backward'):
        op = getattr(torch.ops.xnwan, name)
        assert 'xnwan::' + name in str(op.default._schema)
    assert len(ops.OPS) == 4
    # shape inference without a device (fake tensors): what torch.compile / meta tracing sees
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        X = torch.empty(33, 7, 6)
        u, Y = torch.ops.xnwan.xnode_forward(X, torch.empty(33, dtype=torch.float64), torch.empty(100, dtype=torch.float64), 1, 20, 10, 8, True)
        assert u.shape == (33, 7, 1) and Y.shape == (7, 20, 33) and u.dtype == torch.float64
        v = torch.ops.xnwan.testnet_forward(torch.empty(5, 3, 6), torch.empty(100, dtype=torch.float64), 50, 9)
        assert v.shape == (5, 3, 1)


def test_cube_weight_gradient_in_closed_form_equals_autograd():
    """Hypercube.func_w_grad: what torch.autograd.grad(func_w(x).sum(), x) returns, entry for entry -- including ties
    (equal distances to two faces: torch.minimum splits the gradient), points on a face (|.| has slope 0 at 0) and the
    coordinate torch.min reports when several are equally close"""
    from xnode_wan_pde_solver_amd import sampling
    torch.manual_seed(5)
    dom = sampling.Hypercube([-1, 1], 6, 0, 1, 5)
    x = dom.interior(3000)
    x[0, :, 1] = 0.0                  # equally far from both faces of axis 0 ...
    x[0, :, 2:] = 0.0                 # ... and of every other axis
    x[1, :, 2] = 1.0                  # on a face
    x[2, :, 3] = -1.0
    x[3, :, 1], x[3, :, 2] = 0.5, -0.5        # top of axis 0 as close as bottom of axis 1
    x[4, :, 1:] = 0.25                        # all axes equally close to the top
    xl = x.clone().requires_grad_(True)
    w = dom.func_w(xl)
    gw, = torch.autograd.grad(w.sum(), xl)
    w2, g2 = dom.func_w_grad(x)
    assert torch.equal(w, w2) and torch.equal(gw, g2)
    xb = dom.boundary(500)
    xl = xb.clone().requires_grad_(True)
    gw, = torch.autograd.grad(dom.func_w(xl).sum(), xl)
    assert torch.equal(gw, dom.func_w_grad(xb)[1])


def test_tanh_instruction_sequence_on_the_host():
    """xw_tanh (csrc/xw_common.h) step by step in numpy with a fused multiply-add emulated in extended precision and a
    reciprocal good to 2^-23 like v_rcp_f64: the 29-instruction sequence stays within 4e-16 of tanh over [-100, 100]"""
    import math
    import struct
    ld = np.longdouble

    def fma(a, b, c):
        return np.float64(ld(a) * ld(b) + ld(c))
    rng = np.random.default_rng(0)

    def xw_tanh(x):
        am = min(abs(x), 40.0)
        y = np.float64(-2.0 * am)
        nb = fma(y, 1.4426950408889634, 6755399441055744.0)
        n = np.float64(nb - 6755399441055744.0)
        r = fma(n, -6.93147180369123816490e-01, y)
        r = fma(n, -1.90821492927058770002e-10, r)
        p = np.float64(2.08767569878681e-09)
        for c in (2.505210838544172e-08, 2.755731922398589e-07, 2.7557319223985893e-06, 2.48015873015873e-05, 0.0001984126984126984,
                  0.001388888888888889, 0.008333333333333333, 0.041666666666666664, 0.16666666666666666, 0.5, 1.0, 1.0):
            p = fma(p, r, c)
        lo, hi = struct.unpack('<II', struct.pack('<d', p))
        nlo = struct.unpack('<II', struct.pack('<d', nb))[0]
        e = struct.unpack('<d', struct.pack('<II', lo, (hi + ((nlo << 20) & 0xffffffff)) & 0xffffffff))[0]
        num, den = np.float64(1.0 - e), np.float64(1.0 + e)
        rc = np.float64((1.0 / den) * (1.0 + rng.uniform(-1, 1) * 2.0 ** -23))
        rc = fma(fma(-den, rc, 1.0), rc, rc)
        q = np.float64(num * rc)
        q = fma(fma(-den, q, num), rc, q)
        return math.copysign(float(q), x)
    xs = np.concatenate([rng.uniform(-100, 100, 3000), rng.uniform(-3, 3, 6000), rng.uniform(-1e-3, 1e-3, 500), [0.0, 40.0, -40.0, 39.9, 1e-8]])
    worst = max(abs(xw_tanh(float(x)) - math.tanh(float(x))) for x in xs)
    assert worst < 4e-16, worst
    assert xw_tanh(0.0) == 0.0 and xw_tanh(50.0) == 1.0 and xw_tanh(-50.0) == -1.0


def test_boundary_faces_table_is_the_reference_loop():
    """Hypercube._faces (one scatter) against the reference's 2 d slice assignments (src/dataset.py:265-272), incl. N_b < 2 d"""
    from xnode_wan_pde_solver_amd import sampling
    for d, nb in ((3, 40), (5, 64), (20, 4096), (4, 5), (2, 7)):
        dom = sampling.Hypercube([-1, 2], d, 0, 1, 4)
        x = torch.zeros(nb, d)
        block = int(nb / d / 2)
        cuts = [block * i for i in range(2 * d)] + [nb]
        for axis in range(d):
            x[cuts[2 * axis]:cuts[2 * axis + 1], axis] = dom.top
            x[cuts[2 * axis + 1]:cuts[2 * axis + 2], axis] = dom.bot
        y = torch.zeros(nb, d)
        rows, axis, val = dom._faces(nb)
        y[rows, axis] = val
        assert torch.equal(x, y), (d, nb)


def test_native_uniform_fill_is_torchs_stream():
    """xw_mt19937_uniform_f32 (csrc/xw_hostrng.cpp) behind sampling._uniform_fill: values, order and the generator state left
    behind are those of Tensor.uniform_ on the default CPU generator -- across regeneration boundaries of the 624-word state,
    from any position in it, for ranges that round differently with and without a fused multiply-add"""
    from xnode_wan_pde_solver_amd import sampling
    fill = sampling._UniformFill()
    fill(torch.empty(8), 0.0, 1.0)
    assert fill.mode in (0, 1), 'the native fill did not reproduce torch on this machine (it would silently fall back)'
    torch.manual_seed(99)
    torch.rand(311)                                   # somewhere inside a state block
    for n, lo, hi in ((2048, -1.0, 1.0), (81920, -1.0, 1.0), (4097, 0.25, 7.5), (2671, -3.0, -1.0), (624 * 5, 0.0, 1.0), (100003, -0.1, 0.3)):
        s0 = torch.get_rng_state()
        a = torch.empty(n).uniform_(lo, hi)
        s1 = torch.get_rng_state()
        torch.set_rng_state(s0)
        b = fill(torch.empty(n), lo, hi)
        assert torch.equal(a, b) and torch.equal(s1, torch.get_rng_state()), (n, lo, hi)
        torch.randn(3)                                # the normal cache in the state blob is carried through untouched
    # small, non-float32 and non-contiguous tensors take torch's own path -- same stream either way
    s0 = torch.get_rng_state()
    a = [torch.empty(10).uniform_(0, 1), torch.empty(5000, dtype=torch.float64).uniform_(0, 1)]
    torch.set_rng_state(s0)
    b = [fill(torch.empty(10), 0, 1), fill(torch.empty(5000, dtype=torch.float64), 0, 1)]
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    # a blob that is not a generator state is refused
    from xnode_wan_pde_solver_amd._lib import lib
    junk = torch.zeros(5056, dtype=torch.uint8)
    assert lib.xw_mt19937_uniform_f32(junk.data_ptr(), junk.numel(), torch.empty(4).data_ptr(), 4, 0.0, 1.0, 1) < 0
    assert lib.xw_mt19937_uniform_f32(s0.data_ptr(), 100, torch.empty(4).data_ptr(), 4, 0.0, 1.0, 1) < 0


def test_native_normal_fill_is_numpys_legacy_stream():
    """xw_mt19937_legacy_normal_f64 (csrc/xw_hostrng.cpp) behind sampling._legacy_normal: values, order and the state left behind
    (key, position, cached second value of a pair) are those of np.random.normal on numpy's global generator -- across state
    blocks, from any position, with and without a cached value going in and coming out; the ball domains' samples follow"""
    import ctypes
    import numpy as np
    from xnode_wan_pde_solver_amd import sampling
    draw = sampling._LegacyNormal()
    draw((2, 2))
    assert draw.ok is True, 'the native fill did not reproduce numpy on this machine (it would fall back, with a warning)'
    keep = np.random.get_state()
    try:
        rs = np.random.RandomState(5)
        sizes = [(10, 8192), 1025, 1, 2048, (3, 1001), 4096, 7, (10, 3), 100001, 1024, 2, 2047 + 1024] + [int(n) for n in rs.randint(1024, 9000, 40)]
        for seed in (0, 31337):
            np.random.seed(seed)
            np.random.rand(211)                               # somewhere inside a state block
            a = [np.random.normal(size=n) for n in sizes]
            ra, sa = np.random.rand(5), np.random.get_state()
            np.random.seed(seed)
            np.random.rand(211)
            b = [draw(n) for n in sizes]
            rb, sb = np.random.rand(5), np.random.get_state()
            assert all(x.shape == y.shape and np.array_equal(x, y) for x, y in zip(a, b))
            assert np.array_equal(ra, rb) and np.array_equal(sa[1], sb[1]) and sa[2:] == sb[2:]
        # the samplers built on it: same groups either way
        for cls, args in ((sampling.NSphere_THourglass, (0.7, 5, 0.0, 1.0, 9)), (sampling.NSphere_TCone, (1.0, 4, 0.0, 1.0, 7))):
            out = []
            for native in (True, False):
                sampling._legacy_normal.ok = None if native else False
                torch.manual_seed(3)
                np.random.seed(3)
                dom = cls(*args)
                out.append((dom.interior(3000), dom.boundary(2000)))
            sampling._legacy_normal.ok = None
            for ga, gb in zip(out[0], out[1]):
                assert len(ga) == len(gb) and all(torch.equal(x, y) for x, y in zip(ga, gb))
        # the diagnostic's sample (interior only): same interior groups, both generators end where the full sample leaves them
        for cls, args in ((sampling.NSphere_THourglass, (1.0, 5, 0.0, 1.0, 9)), (sampling.NSphere_TCone, (0.7, 4, 0.0, 1.0, 7))):
            out = []
            for slim in (False, True):
                torch.manual_seed(11)
                np.random.seed(11)
                dom = cls(*args)
                ld = sampling.Comb_loader(3000, 2000, dom, 'cpu', interior_only=slim)
                out.append((ld.interioru, ld.interiorv, ld.boundary, torch.rand(3), np.random.rand(3), np.random.normal(size=3)))
            full, slim = out
            assert len(full[0]) == len(slim[0]) and all(torch.equal(x, y) for x, y in zip(full[0], slim[0]))
            assert all(torch.equal(x, y) for x, y in zip(full[1], slim[1])) and len(full[2]) > 0 and slim[2] == []
            assert torch.equal(full[3], slim[3]) and np.array_equal(full[4], slim[4]) and np.array_equal(full[5], slim[5])
        # a position outside the state is refused
        from xnode_wan_pde_solver_amd._lib import lib
        key = np.zeros(624, dtype=np.uint32)
        p, h, c = ctypes.c_int(700), ctypes.c_int(0), ctypes.c_double(0.0)
        buf = np.empty(4)
        assert lib.xw_mt19937_legacy_normal_f64(key.ctypes.data, ctypes.byref(p), ctypes.byref(h), ctypes.byref(c), buf.ctypes.data, 4) < 0
    finally:
        np.random.set_state(keep)


@pytest.mark.parametrize('name', ['NSphere_THourglass', 'NSphere_TCone'])
def test_sampling_process_draws_what_the_sampling_thread_draws(name):
    """sampler_proc.SamplerProcess (a forked child that owns both generator streams while train() runs): the first sample, then
    per outer iteration the diagnostic's sample (interior only) and the next domain + sample, in solver._train.draw_ahead's
    order -- groups, per-group hints and the domains' time grids equal to in-process draws from the same seeds, and the streams
    handed back at the end stand where the in-process ones stand; a second begin / shutdown round works on the same child"""
    import numpy as np
    from xnode_wan_pde_solver_amd import sampling, sampler_proc
    cls = sampling.DOMAINS[name]
    setup = dict(shape_param=1.0, dim=5, T0=0.0, T=1.0, N_t=9)
    N_r, N_b, rounds = 3000, 2000, 3
    make = lambda: cls(setup['shape_param'], setup['dim'], setup['T0'], setup['T'], setup['N_t'])   # noqa: E731
    eq = lambda A, B: len(A) == len(B) and all(torch.equal(a.detach(), b.detach()) for a, b in zip(A, B))   # noqa: E731
    keep_t, keep_n = torch.get_rng_state(), np.random.get_state()
    sp = None
    try:
        for seed in (4, 5):                                   # (two rounds on one child)
            torch.manual_seed(seed)
            np.random.seed(seed)
            dom = make()
            ref = [(dom, sampling.Comb_loader(N_r, N_b, dom, 'cpu'))]
            for k in range(rounds):
                after = sampling.Comb_loader(N_r, N_b, dom, 'cpu', interior_only=True)
                if k == rounds - 1:
                    ref.append((after, None, None))
                else:
                    dom = make()
                    ref.append((after, dom, sampling.Comb_loader(N_r, N_b, dom, 'cpu')))
            end_ref = (torch.rand(3), np.random.rand(3), np.random.normal(size=3))
            torch.manual_seed(seed)
            np.random.seed(seed)
            sp = sp or sampler_proc.SamplerProcess(cls, setup, N_r, N_b)
            sp.begin()
            d0, s0 = sp.first()
            assert eq(s0.interioru, ref[0][1].interioru) and eq(s0.boundary, ref[0][1].boundary) and torch.equal(d0.times, ref[0][0].times)
            assert s0._hints == ref[0][1].pack()[4] and len(s0.interioru) > 1
            cur = d0
            for k in range(rounds):
                after, nd, nxt = sp.submit(None, cur, k == rounds - 1).result()
                r = ref[1 + k]
                assert eq(after.interioru, r[0].interioru) and after.boundary == []
                assert (nd is None) == (r[1] is None)
                if nd is not None:
                    assert torch.equal(nd.times, r[1].times) and eq(nxt.interioru, r[2].interioru) and eq(nxt.boundary, r[2].boundary)
                    assert nxt._hints == r[2].pack()[4] and nd.V() == r[1].V()
                cur = nd
            sp.shutdown()
            end = (torch.rand(3), np.random.rand(3), np.random.normal(size=3))
            assert torch.equal(end[0], end_ref[0]) and np.array_equal(end[1], end_ref[1]) and np.array_equal(end[2], end_ref[2])
        # the child belongs to the process: a second handle (another solver) talks to the same one; it ends with the server
        other = sampler_proc.SamplerProcess(cls, setup, 2 * N_r, N_b)
        assert other.proc.pid == sp.proc.pid and other.proc.is_alive()
        other.begin()
        d1, s1 = other.first()
        other.shutdown()
        assert sum(g.shape[0] for g in s1.interioru if g[0, 0, 0] == 0.0) <= 2 * N_r and len(s1.interioru) > 1
    finally:
        if sp is not None:
            sp.close()
        srv = sampler_proc._SERVER
        if srv is not None:
            srv.close()
            assert not srv.proc.is_alive() and not srv.pools
        torch.set_rng_state(keep_t)
        np.random.set_state(keep_n)


def test_ex43_sines_keep_the_upstream_product_order():
    """configs/Ex4_3_funcs._sines takes the d sines in three tensor operations; the product must stay the coordinate-by-coordinate
    loop of the upstream file (configs/Ex4_3_funcs.py:8-12) bit for bit -- the ball-domain trajectory fixtures hang on it"""
    import math
    import configs.Ex4_3_funcs as F

    def upstream(X, first):
        d = X.shape[-1] - 1
        out = 1
        for i in range(d):
            out = out * torch.sin(math.pi / 2 * X[..., first + i] + math.pi / 2 * i)
        return out
    g = torch.Generator().manual_seed(8)
    for dt in (torch.float64, torch.float32):
        for shape in ((64, 7, 11), (33, 4), (5, 1, 3), (300, 12, 21)):
            X = torch.randn(*shape, generator=g, dtype=dt)
            assert torch.equal(upstream(X, 1), F._sines(X, 1)[0]), (dt, shape)


def test_boundary_face_table_is_shared_by_domain_objects_of_one_shape():
    """the training loop builds a new Hypercube per sample: the face table of the boundary sampler is cached per shape, not per
    object, and a different N_b / dim / box gets its own"""
    from xnode_wan_pde_solver_amd import sampling
    a = sampling.Hypercube([-1, 1], 6, 0, 1, 5)._faces(100)
    b = sampling.Hypercube([-1, 1], 6, 0, 1, 9)._faces(100)
    assert a is b
    c = sampling.Hypercube([-1, 2], 6, 0, 1, 5)._faces(100)
    e = sampling.Hypercube([-1, 1], 6, 0, 1, 5)._faces(101)
    assert c is not a and e is not a and float(c[2][0]) == 2.0 and e[0].shape[0] == 101


@pytest.mark.parametrize('name', ['NSphere_TCone', 'NSphere_THourglass'])
def test_packed_list_sample_is_the_sample_and_its_hints_are_what_the_engine_would_read_back(name):
    """Comb_loader.pack (list domains): the groups as views of ONE flat buffer are the groups, bit for bit; the hints are the
    facts Engine.tabulate_sample / load_group otherwise read back from the device group by group (first times, one shared
    time column, boundary group on the interior group's grid); the triples stop at the shorter list like iteration does,
    the interior views cover every interior group (the diagnostic integrates over all of them)."""
    from xnode_wan_pde_solver_amd import sampling
    torch.manual_seed(3)
    np.random.seed(3)
    dom = sampling.resolve_domain(name)(0.7, 3, 0.0, 1.0, 8)
    ld = sampling.Comb_loader(300, 150, dom, 'cpu')
    gu, gb = ld.interioru, ld.boundary
    n = min(len(gu), len(gb))
    triples, hints = ld.device_groups('cpu')
    assert len(triples) == n == len(hints) == len(list(ld))          # (iterating the loader stops at the shorter list)
    for k, ((x, xv, bx), h) in enumerate(zip(triples, hints)):
        assert torch.equal(x, gu[k].detach()) and xv is x and torch.equal(x, ld.interiorv[k].detach())
        assert torch.equal(bx, gb[k].detach())
        assert h['t0'] == float(gu[k].detach()[0, 0, 0]) and h['tb0'] == float(gb[k].detach()[0, 0, 0])
        assert h['shared_times'] == bool(torch.all(gu[k][:, :, 0] == gu[k][:1, :, 0]))
        assert h['same_grid'] == (gb[k].shape[1] == gu[k].shape[1] and bool(torch.equal(gb[k][0, :, 0].double(), gu[k][0, :, 0].double())))
    views, hall = ld.device_interior('cpu')
    assert len(views) == len(gu) == len(hall) and all(torch.equal(a, b.detach()) for a, b in zip(views, gu))
    if name == 'NSphere_THourglass':       # late-entry groups: every path has its own entry time
        assert any(not h['shared_times'] for h in hall)
    cube = sampling.Comb_loader(16, 16, sampling.Hypercube((-1.0, 1.0), 3, 0.0, 1.0, 4), 'cpu')
    assert cube.device_groups('cpu') is None and cube.device_interior('cpu') is None


def test_rank_local_sampling_keeps_the_diagnostic_sample_rank_local():
    """solver._loader: with several ranks and rank_local_sampling an interior_only request (the L^p diagnostic's sample in
    draw_ahead / _iterate_body) is still this rank's share, a RankCubeLoader -- not the global sample on every rank."""
    class W:
        rank, size = 1, 4
    sol = S.NODE_WAN_solver.__new__(S.NODE_WAN_solver)
    sol.setup = {'N_r': 64, 'N_b': 32}
    sol.device = torch.device('cpu')
    sol.device_sampling, sol.tabulate_on_host = False, False
    dom = sampling.Hypercube([-1, 1], 3, 0, 1, 5)
    sol.world, sol.rank_local_sampling = W(), True
    for interior_only in (False, True):
        pts = sol._loader(dom, interior_only=interior_only)
        assert isinstance(pts, sampling.RankCubeLoader) and pts.n_local == 16
    sol.rank_local_sampling = False                       # shared seed: every rank draws the global sample and keeps a slice
    assert isinstance(sol._loader(dom, interior_only=True), sampling.Comb_loader)
    sol.world, sol.rank_local_sampling = None, True       # one GPU: the flag means nothing
    assert isinstance(sol._loader(dom, interior_only=True), sampling.Comb_loader)


def test_engine_options_are_read_from_the_environment_once_and_in_one_place(monkeypatch):
    """options.EngineOptions: defaults, the XW_* mapping of from_env(), what plan() would print -- and that engine.py / solver.py
    read no environment variable themselves"""
    from xnode_wan_pde_solver_amd.options import EngineOptions
    for k in list(os.environ):
        if k.startswith('XW_'):
            monkeypatch.delenv(k)
    monkeypatch.delenv('GPU_MAX_HW_QUEUES', raising=False)
    assert EngineOptions.from_env() == EngineOptions() and EngineOptions().non_default() == {}
    monkeypatch.setenv('XW_GRAPHS', '0')
    monkeypatch.setenv('XW_XPROJ_MIN_D', '7')
    monkeypatch.setenv('XW_ELEMENTWISE_SINGLE_SLICE', '1')
    monkeypatch.setenv('XW_PRIO_DROP_A', '2')
    monkeypatch.setenv('XW_REPLICATE_BELOW', '0')
    o = EngineOptions.from_env()
    assert o.non_default() == {'use_graphs': False, 'xproj_min_d': 7, 'pairwise_single_slice': False, 'prio_drop_A': 2, 'replicate_below': 0}
    pkg = os.path.join(ROOT, 'xnode_wan_pde_solver_amd')
    for name in ('engine.py', 'solver.py'):
        assert 'os.environ' not in open(os.path.join(pkg, name)).read(), name
