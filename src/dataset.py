"""src.dataset of the reference: domain shapes, loader, fillt -- implemented in xnode_wan_pde_solver_amd.sampling."""
from xnode_wan_pde_solver_amd.sampling import *  # noqa: F401,F403
from xnode_wan_pde_solver_amd.sampling import (Hypercube, NSphere_TCone, NSphere_THourglass, Comb_loader, DeviceCubeLoader, fillt,  # noqa: F401
                                                DOMAINS, resolve_domain)
