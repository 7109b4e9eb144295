"""Train the headline configuration (Ex4_1 cube, d=20, N_r=N_b=4096, N_t=32) and log the held-out relative L2 error."""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import configs.Ex4_1_funcs as P
from bench import workload_params
from src.training import NODE_WAN_solver
from utils.auxillary_funcs import rel_err
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
torch.manual_seed(0)
S = NODE_WAN_solver(dict(workload_params(20, 4096, 4096, 32), iterations=50), P.func_a, P.func_b, P.func_c, P.func_h, P.func_f, P.func_g,
                    torch.device('cuda'), './', func_u_sol=P.func_u_sol, p=2)
g = torch.Generator().manual_seed(12345)
hold = S.domain([-1, 1], 20, 0, 1, 32)
Xh = torch.cat((hold.times.view(1, 32, 1).expand(16384, 32, 1), (torch.rand(16384, 1, 20, generator=g) * 2 - 1).expand(16384, 32, 20)), 2).cuda()
os.makedirs('/tmp/tr', exist_ok=True); os.chdir('/tmp/tr')
log, t0, done = [], time.time(), 0
while done < iters:
    S.train(report=False)
    done += S.iterations
    log.append({'outer_iterations': done, 'generator_steps': 2 * done, 'wall_s': round(time.time() - t0, 2),
                'rel_l2_heldout_16384': float(rel_err(Xh, S.u_net, P.func_u_sol, 2, hold.V(), 16384)), 'loss_u': S.last_loss_u})
    print(log[-1], flush=True)
json.dump({'workload': 'Ex4_1 cube d=20 N_r=N_b=4096 N_t=32, seed 0, YAML hyper-parameters, float64', 'log': log},
          open(os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'gpurun_out', 'train_cfg2.json'), 'w'), indent=1)
