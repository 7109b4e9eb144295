"""The flow of the reference's example.ipynb as a script, against this repository's packages.

Same imports (cell 0), the same two dictionaries merged into `params` without a `shape_param` key (cell 10), user
callables with the notebook's signatures, `NODE_WAN_solver(...).train(report=True, report_it=..., show_plt=...)`
(cell 11), then the relative error on a fresh sample.  The PDE is the notebook's:  u_t - laplace(u) - u^2 = f  on the
cube [-1,1]^d x [0,1]  with a product-of-sines solution.

    python examples/notebook_flow.py [iterations] [N_r]
"""
import math
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))

import torch  # noqa: E402

from NODE_WAN_model.training import NODE_WAN_solver  # noqa: E402
from NODE_WAN_model.dataset import *  # noqa: E402,F401,F403  (the notebook imports the domain classes this way)
from utils.auxillary_funcs import rel_err  # noqa: E402

device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')

setup = {'dim': 5, 'N_t': 20, 'N_r': 400, 'N_b': 400, 'T0': 0, 'T': 1}
config = {'alpha': 1e4 * 400 * 25, 'u_layers': 8, 'u_hidden_dim': 20, 'u_hidden_hidden_dim': 10, 'v_layers': 9,
          'v_hidden_dim': 50, 'n1': 2, 'n2': 1, 'u_rate': 0.015, 'v_rate': 0.04, 'min_steps': 5, 'adjoint': False,
          'solver': 'midpoint'}


def _sines(X):
    s = 1
    for i in range(setup['dim']):
        s = s * torch.sin(math.pi / 2 * X[..., i + 1] + math.pi / 2 * i)
    return s


SCALE = (2 / math.pi) ** (-setup['dim'])


def func_u_sol(X):
    return SCALE * 2 * _sines(X) * torch.exp(-X[:, :, 0])


def func_f(X):
    s = _sines(X)
    return SCALE * (math.pi ** 2 - 2) * s * torch.exp(-X[:, :, 0]) - 4 * s ** 2 * torch.exp(-2 * X[:, :, 0])


def func_g(BX):
    return func_u_sol(BX)


def func_h(X):
    return SCALE * 2 * _sines(X)


def func_c(X, y_output_u):
    return -y_output_u


def func_a(X, i, j):
    return torch.ones(X.shape[:-1]) if i == j else torch.zeros(X.shape[:-1])


def func_b(X, i):
    return torch.zeros(X.shape[:-1])


def main(iterations=200, n_r=None, report_it=100, show_plt=False, workdir=None):
    if n_r is not None:
        setup['N_r'] = setup['N_b'] = int(n_r)
    if workdir is not None:
        os.makedirs(workdir, exist_ok=True)
        os.chdir(workdir)                       # the side files (losses/L2/Time json, best weights, plots) land in cwd
    params = {**config, **setup, **{'iterations': int(iterations)}, **{'domain': 'Hypercube'}}
    solver = NODE_WAN_solver(params, func_a, func_b, func_c, func_h, func_f, func_g, device, './', func_u_sol=func_u_sol, p=2)
    solver.train(report=True, report_it=report_it, show_plt=show_plt)
    domain = Hypercube([-1, 1], setup['dim'], setup['T0'], setup['T'], setup['N_t'])  # noqa: F405
    X = domain.interior(4096)
    err = float(rel_err(X, solver.u_net, func_u_sol, 2, domain.V(), 4096))
    print('relative L2 error on a fresh sample of 4096 paths: %.4f' % err)
    return solver, err


if __name__ == '__main__':
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 200, sys.argv[2] if len(sys.argv) > 2 else None)
