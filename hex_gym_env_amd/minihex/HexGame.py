"""alias of hex_gym_env_amd.hex_game (reference module name: minihex/HexGame.py)"""
from ..hex_game import HexEnv, HexGame, player, random_policy  # noqa: F401
