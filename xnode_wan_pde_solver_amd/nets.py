"""Network shells: nn.Modules with the reference's attribute names / state_dict keys whose parameters are VIEWS into
one flat float64 blob per network -- the blob is what the HIP kernels read and what the fused Adam kernel updates.

Call surface mirrored (file:line into the reference):
    XNODE            <- NeuralODE        src/model.py:54-112      u_theta: lift h/g -> hidden state, fixed-grid ODE, read-out
    HiddenField      <- _ODEField        src/model.py:115-141     the field MLP (one weight-tied hidden Linear)
    TestNet          <- discriminator    src/model.py:18-51       v_phi (one weight-tied hidden Linear)
    PathParallel     <- nn.DataParallel  src/training.py:93-97    only the `.module` wrapper / `module.` key prefix; the
                                                                  actual multi-GPU strategy is in dist.py
    init_weights                          src/model.py:12-15
Module construction order and `apply(init_weights)` order are the reference's, so seeded runs draw identical weights
(pinned by tests/golden).

The modules' forward()/backward run on the HIP kernels through two autograd Functions, so `u_net(X)`, `v_net(XV)`,
`.backward()` and torch optimisers keep working for user code; the training loop itself does not use autograd at all
(engine.py).
"""
import torch
from torch import nn

from . import kernels as KN
from . import ops as _ops  # noqa: F401  (registers the xnwan:: operators)
from ._lib import XnwanError

F64 = torch.float64


def init_weights(layer):
    """Xavier-uniform weights, zero bias on every Linear (reference src/model.py:12-15)."""
    if type(layer) == nn.Linear:
        nn.init.xavier_uniform_(layer.weight)
        layer.bias.data.fill_(0)


# ----------------------------------------------------------------------------------------------------------------------
# flat parameter storage
# ----------------------------------------------------------------------------------------------------------------------
class Blob:
    """One contiguous float64 vector holding all parameters of a network in the kernels' layout (include/xnwan.h:
    named_parameters() order, row-major matrices) -- at the width of the kernel instantiation the network runs in.

    `slots`: for every parameter of named_parameters() its place in the blob, (offset, rows, cols, leading dimension[,
    column offset]); missing slots (a field without hidden layer, u_layers = 1) stay zero.  When the network is as wide as
    the instantiation the parameters are contiguous pieces; a narrower network is EMBEDDED: every matrix is the leading
    block of its zero-padded container (the y-columns of Win keep their place behind x and t), so the module's parameters
    become strided views and the padding never leaves zero (kernels.ODE_WIDTHS)."""

    def __init__(self, module, device, slots=None, total=None):
        params = [p for _, p in module.named_parameters()]      # de-duplicated: tied layers appear once
        self.names = [n for n, _ in module.named_parameters()]
        self.shapes = [tuple(p.shape) for p in params]
        self.sizes = [p.numel() for p in params]
        if slots is None:                                       # plain concatenation
            slots, off = [], 0
            for p in params:
                r, c = (p.shape[0], p.shape[1]) if p.dim() == 2 else (1, p.numel())
                slots.append((off, r, c, c))
                off += p.numel()
            total = off
        self.data = torch.zeros(total, dtype=F64, device=device)
        self.grad = torch.zeros_like(self.data)
        self.slots = slots
        self.offsets = [sl[0] for sl in slots]
        for p, (off, r, c, ld) in zip(params, slots):
            if p.dtype != F64:
                raise XnwanError('network parameters must be float64 (the reference computes in float64 throughout)')
            view = self._view(self.data, p.shape, off, ld)
            view.copy_(p.data)
            p.data = view                                       # parameter now aliases the blob
        self.params = params
        self.embedded = any(p.dim() == 2 and sl[3] != p.shape[1] for p, sl in zip(params, slots)) or total != sum(self.sizes)

    @staticmethod
    def _view(flat, shape, off, ld):
        if len(shape) == 2:
            return flat.as_strided(tuple(shape), (ld, 1), flat.storage_offset() + off)     # (as_strided's offset is absolute)
        return flat[off:off + shape[0]]

    def split(self, flat):
        """the per-parameter pieces of a vector laid out like the blob (a gradient), as tensors of the parameters' shapes"""
        return [self._view(flat, s, sl[0], sl[3]) for s, sl in zip(self.shapes, self.slots)]

    def check_alias(self):
        """The kernels read the blob, user code sees the Parameters: make sure they are still the same memory."""
        for p, o in zip(self.params, self.offsets):
            if p.data.data_ptr() != self.data.data_ptr() + 8 * o:
                raise XnwanError('a parameter no longer aliases its blob (module moved/cast after construction?) -- '
                                 'call solver.rebind() after .to()/.double()')


def _u_slots(d, H, K, Hc, Kc, has_hidden):
    """places of u_theta's parameters (named_parameters() order) in the blob of width (Hc, Kc); include/xnwan.h layout"""
    ldin, p = d + 1 + Hc, 0
    sl = []
    for r, c, rc, cc in ((H, 1, Hc, 1), (H, None, Hc, None), (H, H, Hc, Hc), (H, None, Hc, None), (H, H, Hc, Hc), (H, None, Hc, None)):
        sl.append((p, r, c if c else 1, cc if cc else 1))       # IL0.w, IL0.b, IL2.w, IL2.b, IL4.w, IL4.b
        p += rc * (cc if cc else 1)
    sl.append((p, K, d + 1 + H, ldin)); p += Kc * ldin          # Win (x | t | y columns: the y block keeps its offset d + 1)
    sl.append((p, K, 1, 1)); p += Kc                            # Win.b
    if has_hidden:
        sl.append((p, K, K, Kc))
    p += Kc * Kc                                                # Wh (slot kept zero when the field has no hidden layer)
    if has_hidden:
        sl.append((p, K, 1, 1))
    p += Kc
    sl.append((p, H, K, Kc)); p += Hc * Kc                      # Wo
    sl.append((p, H, 1, 1)); p += Hc                            # Wo.b
    sl.append((p, 1, H, Hc)); p += Hc                           # FL.w
    sl.append((p, 1, 1, 1)); p += 1                             # FL.b
    return sl, p


def _v_slots(d, W, Wc):
    p, sl = 0, []
    sl.append((p, W, d + 1, d + 1)); p += Wc * (d + 1)          # Vin
    sl.append((p, W, 1, 1)); p += Wc
    sl.append((p, W, W, Wc)); p += Wc * Wc                      # Vh
    sl.append((p, W, 1, 1)); p += Wc
    sl.append((p, 1, W, Wc)); p += Wc                           # Vo
    sl.append((p, 1, 1, 1)); p += 1
    return sl, p


# ----------------------------------------------------------------------------------------------------------------------
# autograd bridges (compat path for user code; the training loop calls the kernels directly)
# ----------------------------------------------------------------------------------------------------------------------
class _OdeFn(torch.autograd.Function):
    """u_net(X).backward(): both directions are the registered operators xnwan::xnode_forward / xnode_backward (ops.py);
    this Function only routes the flat parameter gradient to the module's parameters (views of the blob)."""

    @staticmethod
    def forward(ctx, X, start, net, *params):
        blob = net.blob
        blob.check_alias()
        need = any(ctx.needs_input_grad)
        u, Y = torch.ops.xnwan.xnode_forward(X, start, blob.data, net.method, net.kdims[0], net.kdims[1], net.num_layers, need)
        ctx.net, ctx.x_dtype, ctx.s_shape, ctx.s_dtype = net, X.dtype, start.shape, start.dtype
        ctx.save_for_backward(X.detach(), start.detach(), Y)
        return u

    @staticmethod
    def backward(ctx, gu):
        net = ctx.net
        X, start, Y = ctx.saved_tensors
        want_p = any(ctx.needs_input_grad[3:])
        gx, gs, gflat = torch.ops.xnwan.xnode_backward(gu.contiguous(), X, start, Y, net.blob.data, net.method, net.kdims[0],
                                                       net.kdims[1], net.num_layers, bool(net.adjoint), want_p)
        gX = None
        if ctx.needs_input_grad[0]:
            # nabla_x u is deposited at time index 0 (the model reads x from slice 0 only, src/model.py:99); the time
            # channel's gradient (non-zero only on path 0 in the reference, never read by the loss) is returned as 0
            gX = torch.zeros(X.shape, dtype=ctx.x_dtype, device=gu.device)
            gX[:, 0, 1:] = gx.to(ctx.x_dtype)
        gS = gs.view(ctx.s_shape).to(ctx.s_dtype) if ctx.needs_input_grad[1] else None
        gp = net.blob.split(gflat) if want_p else [None] * len(net.blob.params)
        return (gX, gS, None) + tuple(gp)


class _DiscFn(torch.autograd.Function):
    """v_net(XV).backward() through xnwan::testnet_forward / testnet_backward (ops.py)"""

    @staticmethod
    def forward(ctx, XV, net, *params):
        net.blob.check_alias()
        ctx.net, ctx.dtype = net, XV.dtype
        ctx.save_for_backward(XV.detach())
        return torch.ops.xnwan.testnet_forward(XV, net.blob.data, net.kwidth, net.num_layers)

    @staticmethod
    def backward(ctx, gv):
        net = ctx.net
        XV, = ctx.saved_tensors
        want_p = any(ctx.needs_input_grad[2:])
        gX, gflat = torch.ops.xnwan.testnet_backward(gv.contiguous(), XV, net.blob.data, net.kwidth, net.num_layers,
                                                     bool(ctx.needs_input_grad[0]), want_p)
        gp = net.blob.split(gflat) if want_p else [None] * len(net.blob.params)
        return (gX.to(ctx.dtype) if ctx.needs_input_grad[0] else None, None) + tuple(gp)


# ----------------------------------------------------------------------------------------------------------------------
# modules
# ----------------------------------------------------------------------------------------------------------------------
class HiddenField(nn.Module):
    """dh/dt = F([x, t, h]): Linear(H+d+1, K) -> (ReLU -> tied Linear(K, K)) x (layers-1) -> Tanh -> Linear(K, H)."""

    def __init__(self, input_dim, setup, num_layers, hidden_dim):
        super().__init__()
        self.input_dim, self.hidden_dim, self.num_layers = input_dim, hidden_dim, num_layers
        if num_layers < 1:
            raise XnwanError('u_layers must be >= 1 (u_layers = 0 is the reference\'s degenerate Linear(H, H-1) field, src/model.py:138)')
        # (u_layers = 1: a field without the tied hidden layer, src/model.py:130; its slot in the blob stays zero)
        tied = [nn.ReLU(), nn.Linear(hidden_dim, hidden_dim)] * (num_layers - 1) if num_layers > 1 else []
        self.net = nn.Sequential(nn.Linear(input_dim + setup['dim'] + 1, hidden_dim), *tied, nn.Tanh(),
                                 nn.Linear(hidden_dim, input_dim)).double()

    def forward(self, h):
        raise XnwanError('the field is only evaluated inside the fused HIP stepper (XNODE.forward)')


class XNODE(nn.Module):
    """u_theta.  forward(inputs[N, L, d+1]) -> [N, L, 1] for a group of equal-length paths (time in channel 0)."""

    def __init__(self, hidden_dim, output_dim, func_h, func_g, setup, hidden_hidden_dim, num_layers, domain,
                 solver='midpoint', min_steps=5, adjoint=False):
        super().__init__()
        if output_dim != 1:
            raise XnwanError('the PDE solution is scalar: output_dim must be 1')
        self.hidden_dim, self.output_dim, self.hidden_hidden_dim = hidden_dim, output_dim, hidden_hidden_dim
        self.h, self.g, self.setup, self.num_layers, self.domain = func_h, func_g, setup, num_layers, domain
        self.solver, self.min_steps, self.adjoint = solver, min_steps, adjoint
        self.method = KN.method_id(solver)
        self.initial_layers = nn.Sequential(nn.Linear(1, hidden_dim), nn.ReLU(), nn.Linear(hidden_dim, hidden_dim),
                                            nn.ReLU(), nn.Linear(hidden_dim, hidden_dim)).double()
        self.ODE_rhs = HiddenField(hidden_dim, setup, num_layers, hidden_hidden_dim)
        self.ODE_rhs.apply(init_weights)
        self.final_linear = nn.Linear(hidden_dim, output_dim).double()
        self.blob = None

    def bind(self, device):
        """move to the device and alias the parameters to the kernels' blob, at the width of the smallest stepper
        instantiation that holds this network (kernels.ode_container; equal widths: plain concatenation)"""
        self.to(device)
        H, K, d = self.hidden_dim, self.hidden_hidden_dim, self.setup['dim']
        self.kdims = KN.ode_container(H, K, self.num_layers)
        slots, total = _u_slots(d, H, K, self.kdims[0], self.kdims[1], self.num_layers > 1)
        assert total == KN.theta_size(d, *self.kdims)
        self.blob = Blob(self, device, slots, total)
        return self.blob

    def start_values(self, inputs, starts_at_T0=None):
        """h(x) for groups that start at T0, g(t0, x) for groups that start on the boundary (src/model.py:95-96);
        evaluated on the device `inputs` lives on (host tensors -> bitwise the reference's CPU values)."""
        first = inputs[:, 0, :]
        if starts_at_T0 is None:
            starts_at_T0 = float(inputs[0, 0, 0].detach()) == self.setup['T0']
        if starts_at_T0:
            return self.h(first).reshape(-1).double()
        return self.g(first.unsqueeze(1)).reshape(-1).double()

    def forward(self, inputs, starts_at_T0=None):
        """starts_at_T0 (optional): the caller knows where the paths start -- True: at T0 (start values h), False: on the
        moving boundary (start values g; the sampler's late-entry groups).  Saves reading inputs[0, 0, 0] / max(w) back from the
        device, which waits for everything queued on it (the training loop's diagnostic passes it).  None: looked up, and
        paths that start neither at T0 nor on the boundary take the evaluation path (bound_pad)."""
        if self.blob is None:
            raise XnwanError('XNODE.bind(device) has not been called')
        served = self.__dict__.get('_served')
        if served is not None and inputs is served[0] and not torch.is_grad_enabled():
            # the training loop's own sample, asked for by a `stop` hook (solver._stop_agreed): the group's device buffers already
            # hold the points, the grid and the start values -- the stepper's forward kernel is launched on them directly
            return served[1]()
        dev = self.blob.data.device
        known = starts_at_T0 is not None
        if starts_at_T0 is None:
            starts_at_T0 = float(inputs[0, 0, 0].detach()) == self.setup['T0']
        gather = None
        if not starts_at_T0 and not known:
            on_boundary = float(torch.max(self.domain.func_w(inputs[:, 0, :].detach().unsqueeze(1)))) < 1e-5
            if not on_boundary:
                # evaluation of points that are neither at T0 nor on the boundary (src/model.py:92-106): integrate from T0
                # over the densified grid of domain.bound_pad / fillt and keep the states at the requested times
                path_i, gather, filled = self.domain.bound_pad(inputs.detach())
                if path_i is not None:
                    # per-bucket grids (hourglass): like the reference, ALL paths are integrated over the bucket's grid
                    # and the bucket's rows / columns are kept; the result is ordered bucket by bucket (src/model.py:104-109)
                    start = self.start_values(inputs)
                    outs = []
                    for rows, cols, grid in zip(path_i, gather, filled):
                        grid = grid.to(inputs.device).to(inputs.dtype).reshape(-1)
                        padded = inputs[:, :1, :].repeat(1, grid.shape[0], 1)
                        padded[:, :, 0] = grid.view(1, -1)
                        out = _OdeFn.apply(padded.to(dev), start.to(dev), self, *self.blob.params)
                        outs.append(out[rows.to(dev)][:, cols.long().to(dev), :])
                    return torch.cat(outs, dim=0)
                grid = filled.to(inputs.device).to(inputs.dtype)
                padded = inputs[:, :1, :].repeat(1, grid.shape[0], 1)
                padded[:, :, 0] = grid.view(1, -1)
                start = self.start_values(inputs)
                out = _OdeFn.apply(padded.to(dev), start.to(dev), self, *self.blob.params)
                return out[:, gather.long().to(dev), :]
        out = _OdeFn.apply(inputs.to(dev), self.start_values(inputs, starts_at_T0).to(dev), self, *self.blob.params)
        if inputs.shape[1] == 1 and starts_at_T0:
            return out[:, 0, :]                                   # reference returns [N, 1] here (src/model.py:89-91)
        return out


class TestNet(nn.Module):
    """v_phi: Linear(d+1, W) -> (ReLU -> tied Linear(W, W)) x v_layers -> Tanh -> Linear(W, 1), pointwise on [..., d+1]."""
    __test__ = False

    def __init__(self, config, setup):
        super().__init__()
        self.num_layers, self.hidden_dim = config['v_layers'], config['v_hidden_dim']
        self.input = nn.Linear(setup['dim'] + 1, self.hidden_dim)
        self.hidden = nn.Linear(self.hidden_dim, self.hidden_dim)
        self.output = nn.Linear(self.hidden_dim, 1)
        self.net = nn.Sequential(self.input, *[nn.ReLU(), self.hidden] * self.num_layers, nn.Tanh(), self.output)
        self.net.double()
        self.blob = None

    def bind(self, device):
        self.to(device)
        d = self.input.in_features - 1
        self.kwidth = KN.disc_container(self.hidden_dim)
        slots, total = _v_slots(d, self.hidden_dim, self.kwidth)
        assert total == KN.phi_size(d, self.kwidth)
        self.blob = Blob(self, device, slots, total)
        return self.blob

    def forward(self, XV):
        if self.blob is None:
            raise XnwanError('TestNet.bind(device) has not been called')
        return _DiscFn.apply(XV.to(self.blob.data.device), self, *self.blob.params)


class PathParallel(nn.Module):
    """Stands where the reference has nn.DataParallel: same `.module` attribute and `module.` state_dict prefix.
    Monte-Carlo paths are sharded across GPUs by one process per GPU (dist.py), not by this wrapper."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *a, **kw):
        return self.module(*a, **kw)


# names under which the reference exports these classes (src/model.py)
NeuralODE = XNODE
discriminator = TestNet
_ODEField = HiddenField
