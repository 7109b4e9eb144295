"""src.model of the reference: NeuralODE (XNODE), discriminator, init_weights -- HIP-backed modules from nets.py."""
from xnode_wan_pde_solver_amd.nets import (NeuralODE, discriminator, _ODEField, XNODE, TestNet, HiddenField,  # noqa: F401
                                            PathParallel, init_weights)
from src.dataset import fillt  # noqa: F401
