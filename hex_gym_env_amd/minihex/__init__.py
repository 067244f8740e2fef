"""Alias package with the reference's module names, so that a script written against
`minihex` only changes its import prefix:

    from hex_gym_env_amd.minihex.HexSingleGame import HexEnv
    from hex_gym_env_amd.minihex.SelfplayWrapper import selfplay_wrapper
    from hex_gym_env_amd import minihex;  minihex.random_policy(board)

The implementations live in hex_gym_env_amd.hex_game / hex_single_game / selfplay_wrapper.
`hex-v0` is registered with gymnasium when gymnasium is installed (minihex/__init__.py:15-18).
"""
from ..hex_game import random_policy  # noqa: F401
from .._compat import HAVE_GYMNASIUM

if HAVE_GYMNASIUM:  # pragma: no cover - gymnasium is absent from the build image
    from gymnasium.envs.registration import register, registry
    if "hex-v0" not in registry:
        register(id="hex-v0", entry_point="hex_gym_env_amd.hex_game:HexEnv")
