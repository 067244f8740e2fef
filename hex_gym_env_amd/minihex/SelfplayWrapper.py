"""alias of hex_gym_env_amd.selfplay_wrapper (reference module name: minihex/SelfplayWrapper.py)"""
from ..selfplay_wrapper import BaseRandomPolicy, OpponentPolicy, player, selfplay_wrapper  # noqa: F401
