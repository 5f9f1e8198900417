"""CPU oracle (test infrastructure only -- see qt_oracle.py header)."""
from .qt_oracle import *  # noqa: F401,F403
from . import qt_oracle  # noqa: F401
