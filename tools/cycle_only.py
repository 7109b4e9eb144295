"""N g,g,d cycles of the headline workload and nothing else (no instrumented pass, no reuse run, no training): the command the
`cycle` section of profiles/rNN_pmc_traffic.json is counted on --
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- python3 tools/cycle_only.py 10
every kernel launch between the two markers belongs to a whole cycle, so (sum of the counter over all launches) / (3 N) is the
HBM traffic of an average sub-step.  The warm-up and the group load are kept apart by name: pmc_summary.py only counts the
kernels of the sub-steps (k_disc_*, k_ode_*, k_weak_*, k_bdry*, k_gen_cots, k_adam, k_slab_sum*) and divides by the launches of
k_disc_rec (one per cycle)."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench as B
import configs.Ex4_1_funcs as P
from src.training import NODE_WAN_solver
from src.dataset import Comb_loader

cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device('cuda', 0)
params = B.workload_params(20, 4096, 4096, 32)
# (other network widths, e.g. the wide containers:  XW_CYCLE_WIDTHS=64,16,128 python3 tools/cycle_only.py 10)
if os.environ.get('XW_CYCLE_WIDTHS'):
    h_, k_, w_ = (int(x) for x in os.environ['XW_CYCLE_WIDTHS'].split(','))
    params.update(u_hidden_dim=h_, u_hidden_hidden_dim=k_, v_hidden_dim=w_)
torch.manual_seed(0)
S = NODE_WAN_solver(params, P.func_a, P.func_b, P.func_c, P.func_h, P.func_f, P.func_g, dev, './', func_u_sol=P.func_u_sol, p=2)
eng, s = S.engine, S.setup
domain = S.domain(s['shape_param'], s['dim'], s['T0'], s['T'], s['N_t'])
torch.manual_seed(1000)
du, dv, bd = Comb_loader(s['N_r'], s['N_b'], domain, dev)[0]
G = eng.load_group(du, dv, bd, domain)
for _ in range(cycles):
    eng.generator_step(G)
    eng.generator_step(G)
    eng.discriminator_step(G)
torch.cuda.synchronize()
print('cycles', cycles, 'loss_v', float(eng.scal[5]))
