"""hex_gym_env_amd -- MI355X-native vectorised Hex environment (libhexvec + host mirror)."""
from . import _lib
from .vec_env import HexVecEnv
from . import dist  # noqa: F401

__all__ = ["HexVecEnv", "_lib"]
