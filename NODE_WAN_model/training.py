from src.training import *  # noqa: F401,F403
from src.training import NODE_WAN_solver, func_eval  # noqa: F401
