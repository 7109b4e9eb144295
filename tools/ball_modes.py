"""The two semantics of a ball domain's single-slice groups, side by side: relative L2 error of u_theta on the fixed multi-slice
probe group of tests/golden/ref_traj_cone_ex43_d3_seed0.npz (Ex4_3, d = 3, N_r = 256, N_b = 128, N_t = 10, seed 0) at every
generator sub-iteration of train().

  default                        : the REFERENCE's arithmetic -- [N,N] pairwise loss terms on the single-slice T0 groups, Adam
                                   skipping the field's parameters there (DESIGN 8 "Q8"); follows the reference's run, which diverges
  XW_ELEMENTWISE_SINGLE_SLICE=1  : the elementwise formulas (what src/loss.py means) on those groups

usage: python tools/ball_modes.py [outer iterations] [alpha]        (run once per mode; prints every 10th value)"""
import json, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import configs.Ex4_3_funcs as F
from src.training import NODE_WAN_solver
z = np.load('tests/golden/ref_traj_cone_ex43_d3_seed0.npz')
params = json.loads(str(z['params_json']))
params.pop('funcs')
params['iterations'] = int(sys.argv[1]) if len(sys.argv) > 1 else 300
if len(sys.argv) > 2:
    params['alpha'] = float(sys.argv[2])
probe, sol = torch.from_numpy(z['probe']), torch.from_numpy(z['probe_sol'])
log = []


def hook(self, pts, domain):
    with torch.no_grad():
        up = self.u_net(probe).squeeze(2).cpu()
    log.append(float(torch.sqrt(torch.mean((up - sol) ** 2) / torch.mean(sol ** 2))))
    return False


torch.manual_seed(int(z['seed'])); np.random.seed(int(z['seed']))
S = NODE_WAN_solver(params, F.func_a, F.func_b, F.func_c, F.func_h, F.func_f, F.func_g, torch.device('cuda'), './',
                    func_u_sol=F.func_u_sol, p=2, stop=hook)
S.tabulate_on_host = True
os.makedirs('/tmp/ball_modes', exist_ok=True); os.chdir('/tmp/ball_modes')
S.train(report=False)
got = np.array(log)
mode = 'elementwise' if os.environ.get('XW_ELEMENTWISE_SINGLE_SLICE', '0') == '1' else 'reference (pairwise)'
print('mode: %s, alpha %g, %d outer iterations, %d logged sub-iterations' % (mode, params['alpha'], params['iterations'], len(got)))
print('rel-L2 on the probe group, every 20th sub-iteration:', ' '.join('%.3g' % v for v in got[::20]))
print('first %.4g   min %.4g (at %d)   last %.4g   median of the last 50: %.4g' % (got[0], got.min(), int(got.argmin()), got[-1], float(np.median(got[-50:]))))
