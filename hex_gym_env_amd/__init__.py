"""hex_gym_env_amd -- MI355X-native vectorised Hex environment (libhexvec + host mirror)."""
from . import _lib
from .vec_env import HexVecEnv

__all__ = ["HexVecEnv", "_lib"]
