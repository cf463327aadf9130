"""STAND-IN (test infrastructure): the repo's own GridConfig under upstream's module name."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
from pogema_amd.grid_config import GridConfig  # noqa: E402,F401
