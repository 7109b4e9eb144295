"""General PDE coefficients for the fixture `ref_general_d4_midpoint` (round 4): a non-identity diffusion tensor a_ij(t, x), a
non-zero advection b_i(t, x) and a non-linear reaction c(u, t, x), in the reference's callable protocol (configs/Ex4_1_funcs.py:
func_a(X, i, j) -> [N, L], func_b(X, i) -> [N, L], func_c(X, u) -> [N, L, 1]).  Imported by make_golden.py (handed to the reference's
own NODE_WAN_solver / func_eval / loss) and by the tests (handed to the oracle and to the engine): the same callables on both sides.
h, f, g, u_sol stay those of Ex4_1 (the data need not be consistent with an exact solution for a one-iteration parity fixture)."""
import torch


def func_a(X, i, j):
    return (1.0 + 0.5 * X[..., 1] ** 2) * (1.0 if i == j else 0.1 * torch.cos(X[..., 2]))


def func_b(X, i):
    return 0.3 * X[..., i + 1] * torch.exp(-X[..., 0])


def func_c(X, y_output_u):
    return -y_output_u ** 2 + torch.sin(X[..., 1:2])
