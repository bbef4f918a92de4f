"""Host-side mirror of the reference's `cglb.backend` package for the CGLB path, backed by libcglb_hip.so."""
from .backend import BACKENDS, Backend, Hip  # noqa: F401
from .config import *  # noqa: F401,F403
