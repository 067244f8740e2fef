"""alias of hex_gym_env_amd.hex_single_game (reference module name: minihex/HexSingleGame.py)"""
from ..hex_single_game import HexEnv, player  # noqa: F401
