"""Diagnostics with the upstream call surface (utils/auxillary_funcs.py:7-30,34-98 of the reference).

`L_norm` / `rel_err` are the "rel-L2 u error" metric of BASELINE.json: Monte-Carlo L^p norms of
u_sol - u_theta over a sample of paths.  They only call `u_net(X)`, so they work with any module that
maps [N, L, d+1] -> [N, L, 1]; with the engine's u_net the forward runs in the HIP stepper kernel.
"""
import numpy as np
import torch


# func_u_sol on the last few samples it was asked about.  The exact solution is PDE data -- a function of the points -- and the
# reference's acceptance rule (configs/Ex4_1_funcs.py:36-37: rel_err after every generator sub-iteration) evaluates it FOUR times
# per outer iteration on the same [N, L, d+1] tensor (error and norm, twice): one evaluation per sample here.  Keyed by the tensor
# OBJECT and its version counter (an in-place change is a miss); the entries hold their tensors, so an id is never reused.
# Scope: the entries are dropped when train() returns (NODE_WAN_solver.train -> clear_exact_cache), so nothing outlives a run; a
# tensor whose MEMORY is rewritten without a version bump (raw kernels writing through data_ptr, graph replays into static buffers)
# must not be handed to L_norm / rel_err twice -- the engine's own callers build a fresh tensor per sample.
_EXACT = []


def clear_exact_cache():
    """forget the cached func_u_sol evaluations (and release the two samples and values they keep alive)"""
    del _EXACT[:]


def _exact(func_u_sol, x):
    if not torch.is_tensor(x) or (x.is_cuda and torch.cuda.is_current_stream_capturing()):
        return func_u_sol(x)                       # (inside a graph capture the value lives in the capture's pool: never kept)
    for ent in _EXACT:
        if ent[0] is x and ent[1] == x._version and ent[2] is func_u_sol:
            return ent[3]
    val = func_u_sol(x)
    _EXACT.insert(0, (x, x._version, func_u_sol, val))
    del _EXACT[2:]
    return val


def _group_mean_p(x, u_net, p, func_u_sol, error):
    target = _exact(func_u_sol, x)
    if error:
        # `.squeeze()` as in the reference (:14,20): it drops EVERY unit axis, so on a single-slice group the prediction [N]
        # meets func_u_sol's [N,1] and the difference is the [N,N] table over all pairs -- reproduced, not "fixed"
        pred = u_net(x).squeeze()
        dev = pred.device
        diff = target.to(dev) - pred
    else:
        diff = target
    return torch.mean(torch.abs(diff) ** p)


def L_norm(X, u_net, p, func_u_sol, volume, N_r, error=True):
    """(volume * mean |u_sol - u_theta|^p)^(1/p); for a list of equal-length groups the group means are
    weighted by their share of the N_r paths (reference :16-21)."""
    with torch.no_grad():
        if not isinstance(X, (list, tuple)):
            return (volume * _group_mean_p(X, u_net, p, func_u_sol, error)) ** (1.0 / p)
        acc = 0.0
        for x in X:
            acc = acc + (x.shape[0] / N_r) * _group_mean_p(x, u_net, p, func_u_sol, error)
        return (volume * acc) ** (1.0 / p)


def rel_err(X, predu, func_u_sol, p, volume, N_r):
    """L^p error relative to the L^p norm of the exact solution on the same sample (reference :25-30)."""
    num = L_norm(X, predu, p, func_u_sol, volume, N_r, error=True)
    den = L_norm(X, predu, p, func_u_sol, volume, N_r, error=False)
    return num / den.to(num.device) if torch.is_tensor(den) else num / den


def proj(u_net, setup, iteration, device, axes=(0, 1), T=1, T0=0, save=False, show=True, resolution=100,
         colours=8, func_u_sol=0):
    """Contour plots of the guess (and of the exact solution / error when known) on a 2-D slice through the
    bounding hypercube, all other coordinates fixed at 0.5 (reference :34-98).  Thin helper around u_net;
    needs matplotlib only when actually called."""
    import matplotlib
    if not show:
        matplotlib.use('Agg')
    import matplotlib.pyplot as plt
    axes = list(axes)
    if len(axes) != 2:
        raise AssertionError('There can only be two axes in the graph to be able to display them')
    sp = setup.get('shape_param', [-1, 1])
    lo, hi = sp if isinstance(sp, (list, tuple)) else (-sp, sp)
    d = setup['dim']
    grid = torch.full((resolution, resolution, d + 1), 0.5)
    xs = torch.linspace(lo, hi, resolution)
    if 0 in axes:
        ts = torch.linspace(T0, T, resolution)
    else:
        ts = torch.linspace(lo, hi, resolution)
        grid[:, :, 0] = T
    m1, m2 = torch.meshgrid(xs, ts, indexing='ij')
    grid[:, :, axes[0]] = m2
    grid[:, :, axes[1]] = m1
    grid = grid.to(device)
    with torch.no_grad():
        guess = u_net(grid).detach().reshape(resolution, resolution).cpu()
    plt.clf()
    if func_u_sol != 0:
        exact = func_u_sol(grid.cpu()).reshape(resolution, resolution)
        err = guess - exact
        np.save('guess_cn.npy', guess.numpy())
        np.save('error_cn.npy', err.numpy())
        fig, ax = plt.subplots(3)
        for a_, field in zip(ax, (exact, guess, err)):
            fig.colorbar(a_.contourf(xs.numpy(), ts.numpy(), field.numpy(), colours), ax=a_)
        ax[0].set_title('Correct Solution, Guess and Error')
    else:
        fig, ax = plt.subplots(1)
        fig.colorbar(ax.contourf(xs.numpy(), ts.numpy(), guess.numpy(), colours), ax=ax)
        ax.set_title('Guess Solution')
    if save:
        plt.savefig('plot_at_' + str(iteration) + '_along_' + str(axes) + '.png')
    if show:
        plt.show()
    plt.close(fig)
