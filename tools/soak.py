"""soak: N g,g,d cycles of the headline workload on a fixed sample; prints a hash of the final parameters (run twice: the
engine is deterministic, so the two lines must be identical) and checks that everything stayed finite"""
import hashlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import configs.Ex4_1_funcs as P
from bench import workload_params
from src.training import NODE_WAN_solver
from src.dataset import Comb_loader
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
torch.manual_seed(0)
params = workload_params(20, 4096, 4096, 32)
if os.environ.get('XW_CYCLE_WIDTHS'):        # other network widths, e.g. the wide containers: XW_CYCLE_WIDTHS=64,16,128
    h_, k_, w_ = (int(x) for x in os.environ['XW_CYCLE_WIDTHS'].split(','))
    params.update(u_hidden_dim=h_, u_hidden_hidden_dim=k_, v_hidden_dim=w_)
S = NODE_WAN_solver(params, P.func_a, P.func_b, P.func_c, P.func_h, P.func_f, P.func_g,
                    torch.device('cuda'), './', func_u_sol=P.func_u_sol, p=2)
eng, s = S.engine, S.setup
domain = S.domain(s['shape_param'], s['dim'], s['T0'], s['T'], s['N_t'])
du, dv, bd = Comb_loader(s['N_r'], s['N_b'], domain, torch.device('cuda'))[0]
G = eng.load_group(du, dv, bd, domain)
for i in range(n):
    eng.generator_step(G); eng.generator_step(G); eng.discriminator_step(G)
torch.cuda.synchronize()
th, ph = eng.theta.data.cpu(), eng.phi.data.cpu()
assert torch.isfinite(th).all() and torch.isfinite(ph).all() and torch.isfinite(eng.scal).all()
print('cycles %d  loss_u %.17g  loss_v %.17g  sha1(theta|phi) %s' % (n, float(eng.scal[4]), float(eng.scal[5]),
      hashlib.sha1(th.numpy().tobytes() + ph.numpy().tobytes()).hexdigest()))
