"""src.dataset of the reference: domain shapes, loader, fillt -- implemented in xnode_wan_pde_solver_amd.sampling."""
from xnode_wan_pde_solver_amd.sampling import *  # noqa: F401,F403
from xnode_wan_pde_solver_amd.sampling import Hypercube, Comb_loader, fillt, DOMAINS, resolve_domain  # noqa: F401
