"""oracle/refspec.py -- CPU restatement of the XNODE-WAN hot path.  TEST INFRASTRUCTURE ONLY.

This file is the checker, not the product: only tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py may import it.  The product path
(xnode_wan_pde_solver_amd/, src/, utils/) never imports anything from oracle/ and fails
loudly when the HIP library is missing.

It restates, in plain float64 PyTorch-CPU with autograd, what the reference computes on
its adversarial-training hot path (file:line citations are into the upstream repo
paulvoliva/XNODE-WAN-PDE-solver @ v1):

    * parameter initialisation and its RNG draw order     src/training.py:88-100, src/model.py:12-15,34-43,78-85,127-138
    * Hypercube Monte-Carlo sampling                      src/dataset.py:240-290,300-322
    * XNODE primal network u_theta                        src/model.py:87-112,140-156
    * fixed-grid ODE stepper                              call site src/model.py:103-106 (torchdiffeq, third party)
    * test network v_phi                                  src/model.py:30-47
    * coefficient tabulation                              src/training.py:13-43
    * weak functional I and the losses                    src/loss.py:46-96
    * generator / discriminator Adam sub-steps            src/training.py:125-138,151-162
    * L^p error diagnostics                               utils/auxillary_funcs.py:7-30

Pinning status.  Every function except the stepper is pinned against vectors produced by
the reference itself (tests/golden/*.npz, written by tests/golden/make_golden.py in the
build container).  THE STEPPER IS "PARITY UNPINNED": its arithmetic lives in
torchdiffeq==0.1.1 (requirements.txt:7), which is neither vendored under the reference
nor installed/installable here; `odeint_fixed` below restates the published fixed-grid
schemes of that package (grid == requested times, times cast to the state dtype, explicit
euler / explicit midpoint / 3/8-rule rk4) and the golden vectors were generated with a
stand-in that uses the same definition.  The same holds, without any fixture, for
config['adjoint'] = True: `_OdeintAdjoint` restates that package's odeint_adjoint.

Semantics are the ones the reference has ON A GPU: helper
backward passes pollute the parameter gradients (Q1); nabla u and nabla phi enter I as
constants and the s2 / s32 terms carry no gradient (Q2); nabla_x u is the per-path sum
over time deposited at time index 0 (Q3); v is evaluated on a second, independent
interior sample (Q4); no stale .grad carry-over between sub-steps (Q5, CPU-only artefact).
"""
import math
from itertools import product

import torch

F64 = torch.float64


# --------------------------------------------------------------------------------------
# hyper-parameters
# --------------------------------------------------------------------------------------
def split_params(params):
    """config / setup split BY KEY (the reference slices positionally, src/training.py:80-83)."""
    cfg_keys = ['alpha', 'u_layers', 'u_hidden_dim', 'u_hidden_hidden_dim', 'v_layers', 'v_hidden_dim', 'n1', 'n2',
                'u_rate', 'v_rate', 'min_steps', 'adjoint', 'solver']
    setup_keys = ['dim', 'N_t', 'N_r', 'N_b', 'T0', 'T', 'shape_param']
    config = {k: params[k] for k in cfg_keys}
    setup = {k: params[k] for k in setup_keys if k in params}
    setup.setdefault('shape_param', [-1, 1])
    return config, setup


# --------------------------------------------------------------------------------------
# parameter initialisation with the reference's RNG draw order
# --------------------------------------------------------------------------------------
def _default_linear_draw(n_out, n_in):
    """RNG consumption of torch.nn.Linear(n_in, n_out).__init__ in float32 (values are discarded later)."""
    w = torch.empty(n_out, n_in)
    torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))
    bound = 1.0 / math.sqrt(n_in)
    torch.empty(n_out).uniform_(-bound, bound)
    return w


def _xavier64(n_out, n_in):
    w = torch.empty(n_out, n_in, dtype=F64)
    torch.nn.init.xavier_uniform_(w)
    return w


def init_parameters(config, setup):
    """Returns (theta, phi): dicts of float64 tensors, after consuming the global torch RNG exactly like
    NODE_WAN_solver.__init__ does AFTER its Hypercube construction (src/training.py:93-100).

    theta keys: IL0_w IL0_b IL2_w IL2_b IL4_w IL4_b Win Win_b Wh Wh_b Wo Wo_b FL_w FL_b
    phi keys:   Vin Vin_b Vh Vh_b Vo Vo_b
    """
    d, H, K, W = setup['dim'], config['u_hidden_dim'], config['u_hidden_hidden_dim'], config['v_hidden_dim']
    m = config['u_layers']
    # NeuralODE.__init__: initial_layers (src/model.py:78)
    _default_linear_draw(H, 1), _default_linear_draw(H, H), _default_linear_draw(H, H)
    # _ODEField.__init__: the tied hidden Linear is created first (src/model.py:127), then in / out (:131-134)
    if m > 1:
        _default_linear_draw(K, K)
    _default_linear_draw(K, H + d + 1), _default_linear_draw(H, K)
    # ODE_rhs.apply(init_weights) (src/model.py:82): children order in, tied, out
    _xavier64(K, H + d + 1)
    if m > 1:
        _xavier64(K, K)
    _xavier64(H, K)
    # final_linear (src/model.py:85)
    _default_linear_draw(1, H)
    # discriminator.__init__ (src/model.py:34-36)
    _default_linear_draw(W, d + 1), _default_linear_draw(W, W), _default_linear_draw(1, W)
    # u_net.apply(init_weights) (src/training.py:99)
    theta = {}
    theta['IL0_w'], theta['IL2_w'], theta['IL4_w'] = _xavier64(H, 1), _xavier64(H, H), _xavier64(H, H)
    theta['Win'] = _xavier64(K, H + d + 1)
    theta['Wh'] = _xavier64(K, K) if m > 1 else torch.zeros(K, K, dtype=F64)
    theta['Wo'] = _xavier64(H, K)
    theta['FL_w'] = _xavier64(1, H)
    for k, n in (('IL0_b', H), ('IL2_b', H), ('IL4_b', H), ('Win_b', K), ('Wh_b', K), ('Wo_b', H), ('FL_b', 1)):
        theta[k] = torch.zeros(n, dtype=F64)
    # v_net.apply(init_weights) (src/training.py:100): input/hidden/output are visited twice (direct + via .net)
    phi = {}
    for _ in range(2):
        phi['Vin'], phi['Vh'], phi['Vo'] = _xavier64(W, d + 1), _xavier64(W, W), _xavier64(1, W)
    for k, n in (('Vin_b', W), ('Vh_b', W), ('Vo_b', 1)):
        phi[k] = torch.zeros(n, dtype=F64)
    return theta, phi


# reference parameter names (named_parameters() order) <-> oracle keys
U_NAME_MAP = [('module.initial_layers.0.weight', 'IL0_w'), ('module.initial_layers.0.bias', 'IL0_b'),
              ('module.initial_layers.2.weight', 'IL2_w'), ('module.initial_layers.2.bias', 'IL2_b'),
              ('module.initial_layers.4.weight', 'IL4_w'), ('module.initial_layers.4.bias', 'IL4_b'),
              ('module.ODE_rhs.net.0.weight', 'Win'), ('module.ODE_rhs.net.0.bias', 'Win_b'),
              ('module.ODE_rhs.net.2.weight', 'Wh'), ('module.ODE_rhs.net.2.bias', 'Wh_b'),
              ('module.ODE_rhs.net.%d.weight', 'Wo'), ('module.ODE_rhs.net.%d.bias', 'Wo_b'),
              ('module.final_linear.weight', 'FL_w'), ('module.final_linear.bias', 'FL_b')]
V_NAME_MAP = [('module.input.weight', 'Vin'), ('module.input.bias', 'Vin_b'),
              ('module.hidden.weight', 'Vh'), ('module.hidden.bias', 'Vh_b'),
              ('module.output.weight', 'Vo'), ('module.output.bias', 'Vo_b')]


def u_names(m):
    last = 2 * m  # index of the output Linear inside _ODEField.net (src/model.py:131-135)
    names = [(a % last if '%d' in a else a, b) for a, b in U_NAME_MAP]
    # u_layers = 1: no tied hidden Linear exists (src/model.py:127-130, `additional_layers = ... if num_layers > 1 else []`) and
    # net.2 IS the output layer; the oracle's Wh / Wh_b stay zero and have no counterpart in the reference's state_dict
    return [nb for nb in names if not (m == 1 and nb[1] in ('Wh', 'Wh_b'))]


# --------------------------------------------------------------------------------------
# Hypercube sampling (src/dataset.py:240-290) and the loader draw order (:300-310)
# --------------------------------------------------------------------------------------
class Cube:
    def __init__(self, shape_param, d, T0, T, N_t):
        self.bot, self.top = shape_param[0], shape_param[1]
        self.d, self.T0, self.T, self.N_t = d, T0, T, N_t
        self.times, _ = torch.sort(torch.Tensor(N_t).uniform_(T0, T), 0)  # :248
        self.times[0], self.times[-1] = T0, T                             # :249

    def _paths(self, x):
        n = x.shape[0]
        t = self.times.view(1, -1, 1).expand(n, -1, 1)
        return torch.cat((t, x.view(n, 1, self.d).expand(-1, self.N_t, -1)), 2).contiguous()

    def interior_x(self, n):
        return torch.Tensor(n, 1, self.d).uniform_(self.bot, self.top).view(n, self.d)  # :252

    def boundary_x(self, n_b):
        x = torch.Tensor(n_b, 1, self.d).uniform_(self.bot, self.top).view(n_b, self.d)  # :258
        torch.Tensor(n_b, 1, self.d).uniform_(self.bot, self.top)                        # :263 (drawn, unused)
        blk = int(n_b / self.d / 2)                                                       # :265
        edges = [blk * i for i in range(2 * self.d)] + [n_b]                              # :266-268
        for i in range(self.d):                                                           # :270-272
            x[edges[2 * i]:edges[2 * i + 1], i] = self.top
            x[edges[2 * i + 1]:edges[2 * i + 2], i] = self.bot
        return x[torch.randperm(n_b)]                                                     # :274-276

    def sample(self, n_r, n_b):
        """(X, XV, BX) in the loader's draw order: interior, interior again for v, boundary (:304-310)."""
        xu = self.interior_x(n_r)
        xv = self.interior_x(n_r)
        xb = self.boundary_x(n_b)
        return self._paths(xu), self._paths(xv), self._paths(xb)

    def func_w(self, X):
        """distance to the nearest face (:278-282); X is [N,L,d+1]."""
        xs = X[:, :, 1:]
        return torch.minimum(torch.min(torch.abs(self.top - xs), dim=2).values,
                             torch.min(torch.abs(self.bot - xs), dim=2).values)

    def V(self):
        return (self.top - self.bot) ** self.d * (self.T - self.T0)  # :289-290


# --------------------------------------------------------------------------------------
# networks
# --------------------------------------------------------------------------------------
def field(theta, m, x64, t, y):
    """F([x, t, y]) (src/model.py:153-156 + :130-141): in-layer, (m-1) tied ReLU layers, tanh, out-layer."""
    z = torch.cat((x64, t.reshape(1, 1).expand(y.shape[0], 1), y), 1) @ theta['Win'].T + theta['Win_b']
    for _ in range(m - 1):
        z = torch.relu(z) @ theta['Wh'].T + theta['Wh_b']
    return torch.tanh(z) @ theta['Wo'].T + theta['Wo_b']


def odeint_fixed(f, y0, t, method):
    """Fixed-grid integration on the requested grid (torchdiffeq fixed-grid solvers; PARITY UNPINNED, see header)."""
    t = t.to(y0.dtype)
    ys, y = [y0], y0
    for l in range(t.shape[0] - 1):
        t0, dt = t[l], t[l + 1] - t[l]
        if method == 'euler':
            inc = dt * f(t0, y)
        elif method == 'midpoint':
            inc = dt * f(t0 + dt / 2, y + f(t0, y) * dt / 2)
        elif method == 'rk4':
            k1 = f(t0, y)
            k2 = f(t0 + dt / 3, y + dt * k1 / 3)
            k3 = f(t0 + 2 * dt / 3, y + dt * (k2 - k1 / 3))
            k4 = f(t0 + dt, y + dt * (k1 - k2 + k3))
            inc = dt * (k1 + 3 * (k2 + k3) + k4) / 8
        else:
            raise ValueError('unsupported fixed-grid solver: %r' % (method,))
        y = y + inc
        ys.append(y)
    return torch.stack(ys, 1)  # [N, L, H]


def _rk_step_tuple(f, t0, dt, ys, method):
    """one fixed-grid step of `method` for a state that is a list of tensors (the same tableaux as odeint_fixed)"""
    def axpy(base, terms):
        return [b_ + sum(c * k[i] for c, k in terms) for i, b_ in enumerate(base)]
    if method == 'euler':
        return axpy(ys, [(dt, f(t0, ys))])
    if method == 'midpoint':
        k1 = f(t0, ys)
        return axpy(ys, [(dt, f(t0 + dt / 2, axpy(ys, [(dt / 2, k1)])))])
    if method == 'rk4':
        k1 = f(t0, ys)
        k2 = f(t0 + dt / 3, axpy(ys, [(dt / 3, k1)]))
        k3 = f(t0 + 2 * dt / 3, axpy(ys, [(-dt / 3, k1), (dt, k2)]))
        k4 = f(t0 + dt, axpy(ys, [(dt, k1), (-dt, k2), (dt, k3)]))
        return axpy(ys, [(dt / 8, k1), (3 * dt / 8, k2), (3 * dt / 8, k3), (dt / 8, k4)])
    raise ValueError('unsupported fixed-grid solver: %r' % (method,))


class _OdeintAdjoint(torch.autograd.Function):
    """torchdiffeq.odeint_adjoint for a fixed-grid method, restated from the published torchdiffeq 0.1.1
    (OdeintAdjointMethod; the package is absent from the reference tree -- PARITY UNPINNED, like odeint_fixed):
    the forward pass keeps no graph; the backward pass integrates, for i = L-1 .. 1, the augmented system
        d/dt (y, a, p) = ( f(t, y), -a^T df/dy, -a^T df/dparams )
    from t_i to t_{i-1} with the same method on the two-point grid [t_i, t_{i-1}] (one step), every interval restarted
    from the forward solution y(t_i), and adds the incoming cotangent of y(t_{i-1}) to `a` after it.
    `make_f(params)` builds the field from the module's parameters: whatever else the field closes over (the sample
    point x of src/model.py:99,146-156) receives no gradient -- it is not among odeint_adjoint's inputs."""

    @staticmethod
    def forward(ctx, y0, t, method, make_f, *params):
        with torch.no_grad():
            ans = odeint_fixed(make_f(params), y0, t, method)
        ctx.method, ctx.make_f = method, make_f
        ctx.save_for_backward(t, ans, *params)
        return ans

    @staticmethod
    def backward(ctx, grad_ans):
        t, ans, *params = ctx.saved_tensors
        t = t.to(ans.dtype)
        make_f, method = ctx.make_f, ctx.method

        def aug(tt, state):
            y, a = state[0], state[1]
            with torch.enable_grad():
                y_ = y.detach().requires_grad_(True)
                p_ = [p.detach().requires_grad_(True) for p in params]
                fe = make_f(p_)(tt, y_)
                vj = torch.autograd.grad(fe, [y_] + p_, -a, allow_unused=True)
            vj = [torch.zeros_like(x) if g is None else g for g, x in zip(vj, [y_] + p_)]
            return [fe.detach()] + vj

        L = ans.shape[1]
        adj_y = grad_ans[:, L - 1].clone()
        adj_p = [torch.zeros_like(p) for p in params]
        with torch.no_grad():
            for i in range(L - 1, 0, -1):
                state = _rk_step_tuple(aug, t[i], t[i - 1] - t[i], [ans[:, i], adj_y] + adj_p, method)
                adj_y = state[1] + grad_ans[:, i - 1]
                adj_p = state[2:]
        return (adj_y, None, None, None) + tuple(adj_p)


FIELD_KEYS = ('Win', 'Win_b', 'Wh', 'Wh_b', 'Wo', 'Wo_b')


def u_net(theta, config, X, start_value):
    """NeuralODE.forward for a group that starts at T0 or on the boundary (src/model.py:92-110).
    X [N,L,d+1] float32 (may require grad); start_value [N] = h(X[:,0,:]) or g(X[:,0,:]); returns [N,L] float64."""
    m = config['u_layers']
    s = start_value.reshape(-1, 1).to(F64)
    y0 = torch.relu(torch.relu(s @ theta['IL0_w'].T + theta['IL0_b']) @ theta['IL2_w'].T + theta['IL2_b']) \
        @ theta['IL4_w'].T + theta['IL4_b']                                    # :97
    if X.shape[1] == 1:
        return (y0 @ theta['FL_w'].T + theta['FL_b'])                          # :89-91 (single-slice group)
    x64 = X[:, 0, 1:].to(F64)                                                  # :99 (x from slice 0 only)
    times = X[0, :, 0]                                                         # :92 (path 0's time column)
    if config.get('adjoint'):                                                  # :103 odeint_adjoint
        xc = x64.detach()

        def make_f(params):
            th = dict(zip(FIELD_KEYS, params))
            return lambda t, y: field(th, m, xc, t, y)
        ys = _OdeintAdjoint.apply(y0, times, config['solver'], make_f, *[theta[k] for k in FIELD_KEYS])
    else:
        ys = odeint_fixed(lambda t, y: field(theta, m, x64, t, y), y0, times, config['solver'])
    return (ys @ theta['FL_w'].T + theta['FL_b']).squeeze(2)                   # :110


def v_net(phi, config, XV):
    """discriminator.forward (src/model.py:37-47); XV [..., d+1] -> [...] float64."""
    a = XV.to(F64) @ phi['Vin'].T + phi['Vin_b']
    for _ in range(config['v_layers']):
        a = torch.relu(a) @ phi['Vh'].T + phi['Vh_b']
    return (torch.tanh(a) @ phi['Vo'].T + phi['Vo_b']).squeeze(-1)


# --------------------------------------------------------------------------------------
# coefficient tabulation (src/training.py:25-43) -- same d^2 / d loops, same dtypes
# --------------------------------------------------------------------------------------
def tabulate(funcs, setup, X, BX, u):
    d = setup['dim']
    Xd, BXd = X.detach(), BX.detach()
    h = funcs['h'](Xd[:, 0, :])
    f = funcs['f'](Xd)
    g = funcs['g'](BXd)
    c = funcs['c'](Xd, u.unsqueeze(2))           # attached to u: differentiated through (:29)
    a = torch.empty(d, d, X.shape[0], X.shape[1])
    for i, j in product(range(d), repeat=2):
        a[i, j] = funcs['a'](Xd, i, j)
    b = torch.empty(d, X.shape[0], X.shape[1])
    for i in range(d):
        b[i] = funcs['b'](Xd, i)
    return h, f, g, a, b, c.squeeze(2)


# --------------------------------------------------------------------------------------
# weak functional and losses (src/loss.py:46-96)
# --------------------------------------------------------------------------------------
def weak_I(setup, V, u, v, w, du, dphi, h, f, a, b, c, n_glob=None):
    """I = <A[u], phi> as the reference evaluates it.  du/dphi are CONSTANTS ([N,L,d+1], from the helper
    backward passes); the u factor of s2 and the phi factor of s32 carry no gradient (Q2).
    n_glob: number of paths the 1/N factors refer to when u, v, ... are only one shard of the batch (multi-GPU tests)."""
    d = setup['dim']
    N, L = u.shape
    if n_glob is not None:
        N = n_glob
    phi = v * w                                                               # :52
    s1 = V * (u[:, -1] * v[:, -1] - h * v[:, 0]) / N                          # :64  (v, not phi)
    s2 = V * (u.detach() * dphi[:, :, 0]) / N / L                             # :65  (grad wrt u killed by :75)
    s31 = torch.stack([a[i, j] * dphi[:, :, i + 1] * du[:, :, j + 1]
                       for i, j in product(range(d), repeat=2)], 0).sum(0)   # :66-68
    s32 = sum(b[i] * phi.detach() * du[:, :, i + 1] for i in range(d))        # :69  (grad killed by :74)
    s3 = (V / N / L) * (s31 + s32 + c * u * phi + f * phi)                    # :70-72 (+f*phi as written)
    return torch.sum(s1 - torch.sum(s2 - s3, 1), 0)                           # :73


def interior_loss(V, I, v):
    return torch.log(I ** 2) - torch.log(V * torch.sum(v ** 2) / v.numel())  # :89-90


def init_loss(u, h):
    return torch.mean((u[:, 0] - h) ** 2)                                     # :79


def bdry_loss(u_b, g):
    return torch.mean((u_b - g) ** 2)                                         # :84


# --------------------------------------------------------------------------------------
# one optimiser sub-step, gradients "as Adam sees them"
# --------------------------------------------------------------------------------------
def _leaves(p):
    return {k: t.detach().clone().requires_grad_(True) for k, t in p.items()}


def forward_all(theta, phi, config, setup, cube, funcs, X, XV, BX, need_boundary):
    """Everything both sub-steps share: net outputs, helper-backward quantities, tabulated coefficients, I."""
    V = cube.V()
    Xl = X.detach().clone().requires_grad_(True)
    XVl = XV.detach().clone().requires_grad_(True)
    v = v_net(phi, config, XVl)                                               # src/training.py:129
    u = u_net(theta, config, Xl, funcs['h'](Xl[:, 0, :]))                     # :130
    h, f, g, a, b, c = tabulate(funcs, setup, X, BX, u)                       # :131-133
    w = cube.func_w(XVl)
    # helper backward passes of loss.I (src/loss.py:55,60): d(sum u)/dX, d(sum phi)/dXV
    # and their side effect on the parameter gradients (Q1)
    th_keys = [k for k, p in theta.items() if p.requires_grad]
    ph_keys = [k for k, p in phi.items() if p.requires_grad]
    gu = torch.autograd.grad(u.sum(), [Xl] + [theta[k] for k in th_keys], retain_graph=True, allow_unused=True)
    gp = torch.autograd.grad((v * w).sum(), [XVl] + [phi[k] for k in ph_keys], retain_graph=True, allow_unused=True)
    du, dphi = gu[0], gp[0]                                                    # float32, like X.grad / XV.grad
    pol_theta = {k: (t if t is not None else torch.zeros_like(theta[k])) for k, t in zip(th_keys, gu[1:])}
    pol_phi = {k: (t if t is not None else torch.zeros_like(phi[k])) for k, t in zip(ph_keys, gp[1:])}
    I = weak_I(setup, V, u, v, w.detach(), du, dphi, h, f, a, b, c)
    out = dict(u=u, v=v, w=w.detach(), du=du, dphi=dphi, h=h, f=f, g=g, I=I, V=V,
               pol_theta=pol_theta, pol_phi=pol_phi)
    if need_boundary:
        out['u_b'] = u_net(theta, config, BX.detach(), funcs['h'](BX.detach()[:, 0, :]))  # src/loss.py:84
    return out


def adam_update(p, g, state, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam defaults (src/training.py:103-104)."""
    state['t'] = state.get('t', 0) + 1
    t = state['t']
    for k in p:
        mk = state.setdefault('m_' + k, torch.zeros_like(p[k]))
        vk = state.setdefault('v_' + k, torch.zeros_like(p[k]))
        mk.mul_(beta1).add_(g[k], alpha=1 - beta1)
        vk.mul_(beta2).addcmul_(g[k], g[k], value=1 - beta2)
        denom = (vk.sqrt() / math.sqrt(1 - beta2 ** t)).add_(eps)
        p[k] = p[k] - (lr / (1 - beta1 ** t)) * (mk / denom)
    return p


def generator_grad(theta, phi, config, setup, cube, funcs, X, XV, BX):
    """loss_u and d(loss_u)/d(theta) + pollution, i.e. theta.grad before optimizer_u.step() (src/training.py:127-138)."""
    th = _leaves(theta)
    o = forward_all(th, phi, config, setup, cube, funcs, X, XV, BX, True)
    int_ = interior_loss(o['V'], o['I'], o['v'])
    init_ = init_loss(o['u'], o['h'])
    bdry_ = bdry_loss(o['u_b'], o['g'])
    loss_u = int_ + config['alpha'] * (init_ + bdry_)                         # src/loss.py:93
    gr = torch.autograd.grad(loss_u, list(th.values()), allow_unused=True)
    grad = {k: o['pol_theta'][k] + (g if g is not None else 0) for k, g in zip(th, gr)}
    o.update(int=int_, init=init_, bdry=bdry_, loss=loss_u, grad=grad)
    return o


def discriminator_grad(theta, phi, config, setup, cube, funcs, X, XV, BX):
    """loss_v and phi.grad before optimizer_v.step() (src/training.py:152-162)."""
    ph = _leaves(phi)
    o = forward_all(theta, ph, config, setup, cube, funcs, X, XV, BX, False)
    int_ = interior_loss(o['V'], o['I'], o['v'])
    loss_v = -int_                                                            # src/loss.py:96
    gr = torch.autograd.grad(loss_v, list(ph.values()), allow_unused=True)
    grad = {k: o['pol_phi'][k] + (g if g is not None else 0) for k, g in zip(ph, gr)}
    o.update(int=int_, loss=loss_v, grad=grad)
    return o


# --------------------------------------------------------------------------------------
# list domains (time-varying balls, src/dataset.py:48-229): the group loop of src/training.py:127-138,152-162
# --------------------------------------------------------------------------------------
class Ball:
    """func_w and V of NSphere_TCone (src/dataset.py:199-201,225-229) / NSphere_THourglass (:110-117,154-159).
    Sampling itself is not restated here: the groups come from the fixtures (tests/golden/ref_*_groups.npz)."""

    def __init__(self, kind, r, d, T0, T):
        assert kind in ('NSphere_TCone', 'NSphere_THourglass')
        self.kind, self.r, self.d, self.T0, self.T = kind, r, d, T0, T

    def radius(self, t):
        if self.kind == 'NSphere_TCone':
            return self.r * (1 - t)                                            # :201
        half = (self.T - self.T0) / 2
        return torch.where(t <= half, self.r * ((self.T - self.T0) - t), self.r * t)   # :113-116

    def func_w(self, X):
        return self.radius(X[:, :, 0]) - torch.sqrt(torch.sum(X[:, :, 1:] ** 2, 2))

    def V(self):
        d = self.d
        unit = math.pi ** (d / 2) / math.gamma(d / 2 + 1) * self.r ** d
        if self.kind == 'NSphere_TCone':
            tc = (1 - self.T0) ** (d + 1) / (d + 1) - (1 - self.T) ** (d + 1) / (d + 1)       # :227
        else:
            tc = 2 * ((1 - self.T0) ** (d + 1) / (d + 1) - (1 - (self.T - self.T0) / 2) ** (d + 1) / (d + 1))   # :157-158
        return unit * tc


def u_net_shaped(theta, config, setup, funcs, X):
    """NeuralODE.forward WITH THE SHAPE THE REFERENCE RETURNS (src/model.py:87-112): [N,1] for a single-slice group at
    T0 (the early return of :89-91 skips the trailing unsqueeze), [N,L,1] otherwise.  Start value: h on groups that start
    at T0, g on groups that start on the moving boundary (:95-96)."""
    at_T0 = float(X[0, 0, 0].detach()) == setup['T0']
    start = funcs['h'](X[:, 0, :]) if at_T0 else funcs['g'](X[:, 0, :].unsqueeze(1)).reshape(-1)
    u = u_net(theta, config, X, start)                                        # [N, L]
    return u if (X.shape[1] == 1 and at_T0) else u.unsqueeze(2)


def weak_I_shaped(setup, V, u_s, v3, w, du, dphi, h, f, a, b, c_s):
    """loss.I (src/loss.py:46-76) on tensors that carry the reference's shapes: u_s, c_s as returned by u_net_shaped /
    func_c ([N,1] or [N,L,1]), v3 [N,L,1], w, f [N,L], h [N], du / dphi [N,L,d+1] constants, a [d,d,N,L], b [d,N,L].
    Every `.squeeze()` of the reference drops ALL unit axes, and the products then broadcast: on a single-slice T0 group
    (u_s [N,1] -> [N], against [N,1] operands) s2, s32, c u phi + f phi become [N,N] tables of all PAIRS of paths, summed
    over both axes -- restated literally."""
    d = setup['dim']
    N, L = u_s.shape[0], u_s.shape[1]                                         # :49-50
    phi3 = v3 * w.unsqueeze(2)                                                # :51-52
    uq, vq, pq, cq = u_s.squeeze(), v3.squeeze(), phi3.squeeze(), c_s.squeeze()
    s1 = V * (u_s[:, -1].squeeze() * v3[:, -1].squeeze() - h * v3[:, 0].squeeze()) / N          # :64
    s2 = V * (uq.detach() * dphi[:, :, 0]) / N / L                            # :65 (u-factor: no gradient, Q2)
    s31 = sum(a[i, j] * dphi[:, :, i + 1] * du[:, :, j + 1] for i, j in product(range(d), repeat=2))   # :66-68
    s32 = sum(b[i] * pq.detach() * du[:, :, i + 1] for i in range(d))         # :69 (builtin sum: see make_golden shim 2)
    s3 = (V / N / L) * (s31 + s32 + cq * uq * pq + f * pq)                    # :70-72
    return torch.sum(s1 - torch.sum(s2 - s3, 1), 0)                           # :73


def group_forward(theta, phi, config, setup, domain, funcs, X, XV, BX, need_boundary):
    """one (datau, datav, bdata) triple of the reference's group loop: outputs, helper-backward constants with their
    pollution of the parameter gradients (Q1), tabulated PDE data, I, and the penalties -- reference shapes throughout"""
    V = domain.V()
    Xl = X.detach().clone().requires_grad_(True)
    XVl = XV.detach().clone().requires_grad_(True)
    v3 = v_net(phi, config, XVl).unsqueeze(2)                                 # src/training.py:129
    u_s = u_net_shaped(theta, config, setup, funcs, Xl)                       # :130
    Xd, BXd = X.detach(), BX.detach()
    d = setup['dim']
    h, f, g = funcs['h'](Xd[:, 0, :]), funcs['f'](Xd), funcs['g'](BXd)        # :25-27
    c_s = funcs['c'](Xd, u_s)                                                 # :29
    # (the tables are torch.Tensor(...) = FLOAT32 whatever the sample's dtype, src/training.py:32,37: on the float64 samples of the
    #  ball domains a general a_ij / b_i is rounded to float32 there)
    a = torch.stack([torch.stack([funcs['a'](Xd, i, j) for j in range(d)], 0) for i in range(d)], 0).to(torch.float32)
    b = torch.stack([funcs['b'](Xd, i) for i in range(d)], 0).to(torch.float32)
    w = domain.func_w(XVl)
    th_keys = [k for k, p in theta.items() if p.requires_grad]
    ph_keys = [k for k, p in phi.items() if p.requires_grad]
    gu = torch.autograd.grad(u_s.sum(), [Xl] + [theta[k] for k in th_keys], retain_graph=True, allow_unused=True)   # src/loss.py:55
    gp = torch.autograd.grad((v3 * w.unsqueeze(2)).sum(), [XVl] + [phi[k] for k in ph_keys], retain_graph=True,
                             allow_unused=True)                               # :60
    du, dphi = gu[0], gp[0]
    # None = the helper backward never reached the parameter (the field's, when no ODE step was taken): kept as None,
    # because Adam SKIPS parameters whose .grad is None
    pol_theta = dict(zip(th_keys, gu[1:]))
    pol_phi = dict(zip(ph_keys, gp[1:]))
    I = weak_I_shaped(setup, V, u_s, v3, w.detach(), du, dphi, h, f, a, b, c_s)
    out = dict(u_s=u_s, v3=v3, w=w.detach(), du=du, dphi=dphi, h=h, f=f, g=g, I=I, V=V, pol_theta=pol_theta, pol_phi=pol_phi)
    out['int'] = torch.log(I ** 2) - torch.log(V * torch.sum(v3 ** 2) / (v3.shape[0] * v3.shape[1]))     # :87-90
    if need_boundary:
        out['init'] = torch.mean((u_s[:, 0] - h.unsqueeze(1)) ** 2)           # :79  ([N] - [N,1] -> all pairs on [N,1] outputs)
        ub_s = u_net_shaped(theta, config, setup, funcs, BXd)                 # :84
        out['u_b'] = ub_s
        out['bdry'] = torch.mean((ub_s - g.unsqueeze(2)) ** 2)                # :84  ([n,1] - [n,1,1] -> [n,n,1])
    return out


def _merge_grad(carried, pol, gr):
    """.grad after the helper backward(s) and loss.backward(): None stays None only if nothing ever reached it"""
    out = {}
    for k in gr:
        parts = [t for t in (carried.get(k), pol.get(k), gr[k]) if t is not None]
        out[k] = sum(parts[1:], parts[0]) if parts else None
    return out


def adam_update_sparse(p, g, state, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam with per-parameter step counts; parameters whose gradient is None are skipped entirely (no
    moment decay, no step) -- what torch >= 2.0 does after zero_grad() (set_to_none) for the field of u_theta in the
    first group(s) of a sub-iteration"""
    for k in p:
        if g[k] is None:
            continue
        t = state['t_' + k] = state.get('t_' + k, 0) + 1
        mk = state.setdefault('m_' + k, torch.zeros_like(p[k]))
        vk = state.setdefault('v_' + k, torch.zeros_like(p[k]))
        mk.mul_(beta1).add_(g[k], alpha=1 - beta1)
        vk.mul_(beta2).addcmul_(g[k], g[k], value=1 - beta2)
        denom = (vk.sqrt() / math.sqrt(1 - beta2 ** t)).add_(eps)
        p[k] = p[k] - (lr / (1 - beta1 ** t)) * (mk / denom)
    return p


class GroupLoop:
    """The reference's per-sub-iteration loop over the groups of a list domain (src/training.py:127-138,152-162):
    zero_grad() once (gradients -> None), then for every triple forward + loss + backward + optimizer.step(), gradients
    ACCUMULATING over the groups of the sub-iteration."""

    def __init__(self, theta, phi, config, setup, domain, funcs):
        self.theta, self.phi, self.config, self.setup, self.domain, self.funcs = theta, phi, config, setup, domain, funcs
        self.adam_u, self.adam_v = {}, {}

    def sub_iteration(self, which, triples):
        carried, outs = {}, []
        for (X, XV, BX) in triples:
            if which == 'u':
                th = _leaves(self.theta)
                o = group_forward(th, self.phi, self.config, self.setup, self.domain, self.funcs, X, XV, BX, True)
                loss = o['int'] + self.config['alpha'] * (o['init'] + o['bdry'])                  # src/loss.py:93
                gr = dict(zip(th, torch.autograd.grad(loss, list(th.values()), allow_unused=True)))
                carried = _merge_grad(carried, o['pol_theta'], gr)
                self.theta = adam_update_sparse(self.theta, carried, self.adam_u, self.config['u_rate'])
            else:
                ph = _leaves(self.phi)
                o = group_forward(self.theta, ph, self.config, self.setup, self.domain, self.funcs, X, XV, BX, False)
                loss = -o['int']                                                                  # :96
                gr = dict(zip(ph, torch.autograd.grad(loss, list(ph.values()), allow_unused=True)))
                carried = _merge_grad(carried, o['pol_phi'], gr)
                self.phi = adam_update_sparse(self.phi, carried, self.adam_v, self.config['v_rate'])
            o.update(loss=loss.detach(), grad={k: (None if t is None else t.detach().clone()) for k, t in carried.items()})
            outs.append(o)
        return outs


def l_norm_groups(theta, config, setup, funcs, u_sol, groups, V, p, N_r, error=True):
    """L_norm on a list domain (utils/auxillary_funcs.py:16-23): group-weighted means; `u_net(x).squeeze()` against
    func_u_sol(x) [N,L] -- on a single-slice T0 group [N] against [N,1]: all pairs"""
    diff = 0
    for x in groups:
        with torch.no_grad():
            fx = u_sol(x) - u_net_shaped(theta, config, setup, funcs, x).squeeze() if error else u_sol(x)
        diff = diff + x.shape[0] / N_r * torch.mean(torch.abs(fx) ** p)
    return (V * diff) ** (1 / p)


# --------------------------------------------------------------------------------------
# diagnostics (utils/auxillary_funcs.py:7-30)
# --------------------------------------------------------------------------------------
def l_norm(u_pred, u_true, V, p):
    return (V * torch.mean(torch.abs(u_true - u_pred) ** p)) ** (1 / p)


def rel_err(u_pred, u_true, V, p):
    return l_norm(u_pred, u_true, V, p) / (V * torch.mean(torch.abs(u_true) ** p)) ** (1 / p)


# --------------------------------------------------------------------------------------
# a whole solver, for trajectory-level checks and for the CPU baseline timing
# --------------------------------------------------------------------------------------
class Solver:
    """Same life-cycle as NODE_WAN_solver (src/training.py:65-187) without files, plots or hooks."""

    def __init__(self, params, funcs, u_sol=None, p=2):
        self.config, self.setup = split_params(params)
        self.funcs, self.u_sol, self.p = funcs, u_sol, p
        s = self.setup
        Cube(s['shape_param'], s['dim'], s['T0'], s['T'], s['N_t'])           # src/training.py:88 (RNG only)
        self.theta, self.phi = init_parameters(self.config, s)
        self.adam_u, self.adam_v = {}, {}
        self.rel_log, self.loss_log = [], []

    def new_sample(self):
        s = self.setup
        self.cube = Cube(s['shape_param'], s['dim'], s['T0'], s['T'], s['N_t'])  # :119
        self.X, self.XV, self.BX = self.cube.sample(s['N_r'], s['N_b'])          # :121
        return self.X

    def diagnostic(self, X, relative=False):
        with torch.no_grad():
            u = u_net(self.theta, self.config, X, self.funcs['h'](X[:, 0, :]))
        fn = rel_err if relative else l_norm
        return fn(u, self.u_sol(X), self.cube.V(), self.p).item()

    def generator_step(self):
        o = generator_grad(self.theta, self.phi, self.config, self.setup, self.cube, self.funcs, self.X, self.XV, self.BX)
        self.theta = adam_update(self.theta, o['grad'], self.adam_u, self.config['u_rate'])
        return o

    def discriminator_step(self):
        o = discriminator_grad(self.theta, self.phi, self.config, self.setup, self.cube, self.funcs, self.X, self.XV, self.BX)
        self.phi = adam_update(self.phi, o['grad'], self.adam_v, self.config['v_rate'])
        return o

    def outer_iteration(self, log_rel=True):
        X = self.new_sample()
        if self.u_sol is not None:
            self.diagnostic(X)                                                   # :123
        for _ in range(self.config['n1']):
            o = self.generator_step()
            self.loss_log.append(o['loss'].item())
            if log_rel and self.u_sol is not None:
                self.rel_log.append(self.diagnostic(self.X, relative=True))     # where a `stop` hook would look (:142)
        for _ in range(self.config['n2']):
            self.discriminator_step()
        s = self.setup
        X2, _, _ = self.cube.sample(s['N_r'], s['N_b'])                         # :166 (same domain object, new points)
        if self.u_sol is not None:
            return self.diagnostic(X2)                                           # :167
        return None
