"""src.loss of the reference: the `loss` class (user-facing compat path; the training loop uses the fused kernels)."""
from xnode_wan_pde_solver_amd.compat_loss import loss  # noqa: F401
