"""Command-line entry with the reference's flags (main.py:24-34): solve

    u_t - sum_i d_i( sum_j a_ij d_j u ) + sum_i b_i d_i u + c(u, t, x) = f   in D,   u = g on dD,   u(T0, .) = h

with the XNODE-WAN method on an MI355X.

    python main.py --params cube_pde.yaml --funcs Ex4_1_funcs [-w WORK_DIR] [--device cuda:0] [--report_it 10]
"""
import argparse
import importlib
import os

import torch
import yaml


def parse(argv=None):
    ap = argparse.ArgumentParser(prog='XNODE-WAN PDE solver',
                                 description='a general purpose parabolic PDE solver using the XNODE-WAN architecture (MI355X engine)')
    ap.add_argument('-w', '--work_dir', type=str, default='./', help='directory for the best model parameters')
    ap.add_argument('--params', required=True, help='an experiment setup to load (file under configs/ or a path)')
    ap.add_argument('--funcs', required=True, help='module under configs/ with the functions of the PDE (omit .py)')
    ap.add_argument('--device', default=None, help='device to run on, default cuda')
    ap.add_argument('--report', type=lambda s: str(s).lower() not in ('0', 'false', 'no'), default=True)
    ap.add_argument('--report_it', type=int, default=10, help='number of iterations between reporting progress')
    ap.add_argument('--show_plt', type=lambda s: str(s).lower() in ('1', 'true', 'yes'), default=False)
    ap.add_argument('--iterations', type=int, default=None, help='override the number of outer iterations')
    return ap.parse_args(argv)


def main(argv=None):
    args = parse(argv)
    funcs = importlib.import_module('configs.' + args.funcs)
    here = os.path.dirname(os.path.abspath(__file__))
    path = args.params if os.path.exists(args.params) else os.path.join(here, 'configs', args.params)
    with open(path, 'r') as fh:
        params = yaml.safe_load(fh)
    if args.iterations is not None:
        params['iterations'] = args.iterations
    device = torch.device('cuda') if args.device is None else torch.device(args.device)
    from src.training import NODE_WAN_solver
    solver = NODE_WAN_solver(params, funcs.func_a, funcs.func_b, funcs.func_c, funcs.func_h, funcs.func_f, funcs.func_g,
                             device, args.work_dir, func_u_sol=getattr(funcs, 'func_u_sol', None), p=2,
                             stop=getattr(funcs, 'stop', None))
    solver.train(report=args.report, report_it=args.report_it, show_plt=args.show_plt)


if __name__ == '__main__':
    main()
