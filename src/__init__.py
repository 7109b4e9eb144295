"""Import surface of the reference's `src` package (src/__init__.py:1-4), backed by the MI355X engine."""
from src.dataset import *   # noqa: F401,F403
from src.loss import *      # noqa: F401,F403
from src.model import *     # noqa: F401,F403
from src.training import *  # noqa: F401,F403
