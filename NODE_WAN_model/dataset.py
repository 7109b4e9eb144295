from src.dataset import *  # noqa: F401,F403
from src.dataset import Hypercube, Comb_loader  # noqa: F401
