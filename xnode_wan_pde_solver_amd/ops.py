"""PyTorch custom operators of the hot path (torch.library): what `u_net(X)`, `v_net(XV)` and their `.backward()` dispatch to.

    xnwan::xnode_forward / xnode_backward      <- NeuralODE.forward + its autograd replay   (src/model.py:87-112, src/loss.py:55)
    xnwan::testnet_forward / testnet_backward  <- discriminator.forward + its autograd replay (src/model.py:37-47, src/loss.py:60)

Each operator is a thin shell around the C ABI (include/xnwan.h, through kernels.py): tensors in, tensors out, no Python
state -- the parameters arrive as the flat blob the kernels read plus the list of its views (the module's nn.Parameters), so
that autograd delivers the gradient pieces to the parameters themselves.  The training loop (engine.py) does not go through
autograd at all; these operators serve the reference's module-level call surface (example.ipynb, user code, diagnostics).
"""
from typing import List, Sequence

import torch

from . import kernels as KN

F64 = torch.float64
_LIB = 'xnwan'


# ------------------------------------------------------------------------------------------------------------------------
# u_theta
# ------------------------------------------------------------------------------------------------------------------------
@torch.library.custom_op(_LIB + '::xnode_forward', mutates_args=())
def xnode_forward(X: torch.Tensor, start: torch.Tensor, blob: torch.Tensor, method: int, H: int, K: int, m: int,
                  keep: bool) -> List[torch.Tensor]:
    """X [N, L, d+1] (time in channel 0, x read from time slice 0), start [N] -> [u [N, L, 1], Y [L, H, N] checkpoints]"""
    xT = X[:, 0, 1:].detach().to(F64).t().contiguous()
    t = X[0, :, 0].detach().to(F64).contiguous()
    s = start.detach().to(F64).reshape(-1).contiguous()
    u, Y = KN.ode_fwd(xT, t, s, blob, method, H, K, m, want_Y=keep)
    return [u.t().unsqueeze(2).contiguous(), Y if keep else torch.empty(0, dtype=F64, device=u.device)]


@xnode_forward.register_fake
def _(X, start, blob, method, H, K, m, keep):
    N, L = X.shape[0], X.shape[1]
    return [X.new_empty((N, L, 1), dtype=F64), X.new_empty((L, H, N) if keep else (0,), dtype=F64)]


@torch.library.custom_op(_LIB + '::xnode_backward', mutates_args=())
def xnode_backward(gu: torch.Tensor, X: torch.Tensor, start: torch.Tensor, Y: torch.Tensor, blob: torch.Tensor, method: int,
                   H: int, K: int, m: int, adjoint: bool, want_params: bool) -> List[torch.Tensor]:
    """cotangent gu [N, L, 1] -> [d/dx at time slice 0 [N, d], d/dstart [N], flat parameter gradient (blob layout)]"""
    xT = X[:, 0, 1:].detach().to(F64).t().contiguous()
    t = X[0, :, 0].detach().to(F64).contiguous()
    s = start.detach().to(F64).reshape(-1).contiguous()
    ubar = gu.squeeze(2).t().contiguous().to(F64)
    gx, gs, slab = KN.ode_bwd(xT, t, s, blob, Y, ubar, method, H, K, m, want_x=True, want_params=want_params, adjoint=adjoint)
    gp = KN.slab_sum(slab) if want_params else torch.zeros(0, dtype=F64, device=gu.device)
    return [gx.t().contiguous(), gs, gp]


@xnode_backward.register_fake
def _(gu, X, start, Y, blob, method, H, K, m, adjoint, want_params):
    N, d = X.shape[0], X.shape[2] - 1
    return [X.new_empty((N, d), dtype=F64), X.new_empty((N,), dtype=F64), blob.new_empty(blob.shape if want_params else (0,))]


# ------------------------------------------------------------------------------------------------------------------------
# v_phi
# ------------------------------------------------------------------------------------------------------------------------
@torch.library.custom_op(_LIB + '::testnet_forward', mutates_args=())
def testnet_forward(XV: torch.Tensor, blob: torch.Tensor, W: int, q: int) -> torch.Tensor:
    """XV [..., d+1] points (t, x) -> v [..., 1]"""
    pts = XV.detach().reshape(-1, XV.shape[-1])
    v, _ = KN.disc_fwd(pts[:, 1:].to(F64).t().contiguous(), None, blob, W, q, tpp=pts[:, 0].to(F64).contiguous(), want_vt=False)
    return v.view(XV.shape[:-1]).unsqueeze(-1).contiguous()


@testnet_forward.register_fake
def _(XV, blob, W, q):
    return XV.new_empty(tuple(XV.shape[:-1]) + (1,), dtype=F64)


@torch.library.custom_op(_LIB + '::testnet_backward', mutates_args=())
def testnet_backward(gv: torch.Tensor, XV: torch.Tensor, blob: torch.Tensor, W: int, q: int, want_x: bool,
                     want_params: bool) -> List[torch.Tensor]:
    """cotangent gv [..., 1] -> [input gradient [..., d+1] (time first), flat parameter gradient (blob layout)]"""
    pts = XV.detach().reshape(-1, XV.shape[-1])
    xT, tpp = pts[:, 1:].to(F64).t().contiguous(), pts[:, 0].to(F64).contiguous()
    vbar = gv.reshape(1, -1).contiguous().to(F64)
    gX = torch.zeros(0, dtype=F64, device=gv.device)
    if want_x:
        gxv, gtv = KN.disc_gradx(xT, None, blob, W, q, tpp=tpp, vbar=vbar)
        gX = torch.cat((gtv.view(-1, 1), gxv.t()), 1).view(XV.shape).contiguous()
    gp = KN.slab_sum(KN.disc_bwd(xT, None, blob, vbar, W, q, tpp=tpp)) if want_params else torch.zeros(0, dtype=F64, device=gv.device)
    return [gX, gp]


@testnet_backward.register_fake
def _(gv, XV, blob, W, q, want_x, want_params):
    return [XV.new_empty(XV.shape if want_x else (0,), dtype=F64), blob.new_empty(blob.shape if want_params else (0,))]


OPS = (xnode_forward, xnode_backward, testnet_forward, testnet_backward)
