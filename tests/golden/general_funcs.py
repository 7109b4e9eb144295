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


# ---- round 5: two more coefficient sets, for the special forms the engine stores a table in (Engine._tabulate_a: one [d, d] matrix when a does
# not vary over the sample, its diagonal [d, N] when every off-diagonal entry is exactly zero) and for the linear reaction c = kappa u, with
# which a group's sub-step stays on the fused path (xw_weak_contract_general inside the captured graph / the group runner)
def const_a(X, i, j):
    """one symmetric positive definite matrix for all points"""
    return torch.full(X.shape[:-1], 1.5 if i == j else 0.2 / (1 + abs(i - j)), dtype=X.dtype)


def diag_a(X, i, j):
    """a_ii(x) = 1 + 0.3 x_i^2, every off-diagonal entry exactly zero"""
    return 1.0 + 0.3 * X[..., i + 1] ** 2 if i == j else torch.zeros(X.shape[:-1], dtype=X.dtype)


def zero_b(X, i):
    return torch.zeros(X.shape[:-1], dtype=X.dtype)


def lin_c(X, y_output_u):
    return -0.7 * y_output_u


def variant(name):
    """(func_a, func_b, func_c) of a named set: 'v1' the general one above, 'const' (constant matrix, b = 0, c = -0.7 u), 'diag' (diagonal
    a(x), the general b, c = -0.7 u)"""
    return {'v1': (func_a, func_b, func_c), 'const': (const_a, zero_b, lin_c), 'diag': (diag_a, func_b, lin_c)}[name]
