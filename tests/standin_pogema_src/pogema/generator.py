"""STAND-IN (test infrastructure): upstream's instance generator as recalled, written with numpy itself."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
from oracle.generator_oracle import generate_instance_numpy  # noqa: E402,F401
