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
                                       ('ref_hourglass_ex43_d10_groups', 'NSphere_THourglass')])   # d = 10, Ex4_3 functions
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
