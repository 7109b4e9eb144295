"""src.training of the reference: NODE_WAN_solver, func_eval, init_weights -- driving the MI355X engine."""
from xnode_wan_pde_solver_amd.solver import NODE_WAN_solver, func_eval, split_params, FusedAdam  # noqa: F401
from xnode_wan_pde_solver_amd.nets import init_weights  # noqa: F401
from src.dataset import *  # noqa: F401,F403
from src.dataset import Comb_loader  # noqa: F401
from utils.auxillary_funcs import proj, L_norm, rel_err  # noqa: F401
