"""PDE data for the benchmark problem of Section 4.1 of the XNODE-WAN paper (JCP 463, 111233):

    u_t - laplace(u) - u^2 = f   on [-1,1]^d x [0,1],     u = g on the boundary,   u(0,.) = h

with the closed-form solution  u(t,x) = 2 sin(pi x_1 / 2) cos(pi x_2 / 2) exp(-t)  (only x_1, x_2 enter, so the
same data is valid in any dimension d >= 2).  Same callable protocol as the upstream configs/Ex4_1_funcs.py:5-37:
tensors are [N, L, d+1] with time in channel 0; func_h receives the [N, d+1] slice at the initial time.
"""
import math

import torch

from utils.auxillary_funcs import rel_err

_HALF_PI = math.pi / 2


def _profile(x1, x2):
    return torch.sin(_HALF_PI * x1) * torch.cos(_HALF_PI * x2)


def func_u_sol(X):
    return 2 * _profile(X[..., 1], X[..., 2]) * torch.exp(-X[..., 0])


def func_f(X):
    sc = _profile(X[..., 1], X[..., 2])
    t = X[..., 0]
    return (math.pi ** 2 - 2) * sc * torch.exp(-t) - 4 * sc ** 2 * torch.exp(-2 * t)


def func_g(BX):
    return func_u_sol(BX)


def func_h(X0):
    return 2 * _profile(X0[:, 1], X0[:, 2])


def func_a(X, i, j):
    """diffusion tensor a_ij = delta_ij"""
    shape = X.shape[:-1]
    return torch.ones(shape) if i == j else torch.zeros(shape)


def func_b(X, i):
    """no advection"""
    return torch.zeros(X.shape[:-1])


def func_c(X, y_output_u):
    """reaction term c(u) = -u (differentiated through by the loss)"""
    return -y_output_u


def stop(solver, points, domain):
    """authors' acceptance rule: relative L^p error below one percent"""
    err = rel_err(points, solver.u_net, solver.func_u_sol, solver.p, domain.V(), solver.params['N_r'])
    return bool(err < 0.01)
