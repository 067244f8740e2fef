"""ctypes loader of libhexvec.so (include/hexvec.h).

The product has NO CPU path: importing works anywhere (so the symbol table can
be checked), but every compute entry point needs an MI355X and raises
RuntimeError otherwise.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.environ.get("HEXVEC_SO") or os.path.join(CSRC, "libhexvec.so")  # HEXVEC_SO: A/B experiments

# create flags (include/hexvec.h)
F_ENC_B = 0x1
F_AGENT_WHITE = 0x2
F_AGENT_RANDOM = 0x4
F_NO_AUTO_RESET = 0x8
F_FIRST_WHITE = 0x10
F_TWO_SIDED = 0x20
F_EXACT_LABELS = 0x40
F_ROLLOUT_LAYOUT = 0x80

MIN_BOARD, MAX_BOARD, MAX_BOARD_LABELS = 2, 19, 14

SYMBOLS = [
    "hexvec_create", "hexvec_destroy", "hexvec_reset", "hexvec_step", "hexvec_step_agent",
    "hexvec_step_opponent", "hexvec_apply_moves",
    "hexvec_rollout_random", "hexvec_rollout_selfplay", "hexvec_get_state", "hexvec_set_state", "hexvec_set_boards",
    "hexvec_episode_stats", "hexvec_agent_colours", "hexvec_state_bytes", "hexvec_export_state",
    "hexvec_import_state", "hexvec_stream_synchronize", "hexvec_set_ticket", "hexvec_sample_masked", "hexvec_pick_opponents",
    "hexvec_attach_pool", "hexvec_mlp_lds_bytes",
    "hexvec_mlp_weights_floats", "hexvec_mlp_scratch_bytes", "hexvec_mlp_pack", "hexvec_mlp_actions", "hexvec_board_size", "hexvec_num_envs",
    "hexvec_flags", "hexvec_last_kernel", "hexvec_last_error",
]


def build(force=False):
    """Compile libhexvec.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(_HERE, "..", "include", "hexvec.h"))
    stale = (not os.path.exists(SO_PATH)
             or os.path.getmtime(SO_PATH) < max(os.path.getmtime(f) for f in srcs))
    if force or stale:
        subprocess.check_call(["make", "-s", "-j4", "-C", CSRC, "libhexvec.so"])
    return SO_PATH


_lib = None


def lib():
    """Load the shared library (building it first if the .so is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        try:
            build()
        except Exception as e:  # no hipcc: fail loudly, there is no fallback
            raise RuntimeError(
                f"libhexvec.so is missing and could not be built ({e}); "
                "hex_gym_env_amd has no CPU fallback") from e
    L = C.CDLL(SO_PATH)
    vp, i32, u32, u64 = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64
    L.hexvec_create.argtypes = [C.POINTER(vp), i32, i32, i32, u64, u32, u32]
    L.hexvec_destroy.argtypes = [vp]
    L.hexvec_reset.argtypes = [vp, vp, vp, vp, vp, vp]
    L.hexvec_step.argtypes = [vp] + [vp] * 9
    L.hexvec_step_agent.argtypes = [vp] + [vp] * 9
    L.hexvec_step_opponent.argtypes = [vp] + [vp] * 9
    L.hexvec_apply_moves.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, vp]
    L.hexvec_rollout_random.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp]
    L.hexvec_rollout_selfplay.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp]
    L.hexvec_get_state.argtypes = [vp, vp, vp, vp, vp]
    L.hexvec_set_state.argtypes = [vp, vp, vp, i32, vp]
    L.hexvec_set_boards.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.hexvec_episode_stats.argtypes = [vp, vp, vp]
    L.hexvec_agent_colours.argtypes = [vp, vp, vp]
    L.hexvec_state_bytes.argtypes = [vp]
    L.hexvec_state_bytes.restype = C.c_size_t
    L.hexvec_export_state.argtypes = [vp, vp, vp]
    L.hexvec_import_state.argtypes = [vp, vp, vp]
    L.hexvec_stream_synchronize.argtypes = [vp, vp]
    L.hexvec_set_ticket.argtypes = [vp, vp]
    L.hexvec_sample_masked.argtypes = [vp, C.c_longlong, vp, vp, i32, i32, u64, u32, i32, vp, vp, vp, vp]
    L.hexvec_pick_opponents.argtypes = [vp, i32, vp, vp, vp, vp, i32, C.c_float, i32, u64, u32, i32, vp, vp]
    L.hexvec_attach_pool.argtypes = [vp, vp, vp, C.c_float, i32, u64, vp, vp, vp, i32, vp]
    L.hexvec_mlp_weights_floats.argtypes = [i32, i32, i32]
    L.hexvec_mlp_weights_floats.restype = C.c_size_t
    L.hexvec_mlp_lds_bytes.argtypes = [i32, i32, i32]
    L.hexvec_mlp_lds_bytes.restype = C.c_size_t
    L.hexvec_mlp_scratch_bytes.argtypes = [i32, i32]
    L.hexvec_mlp_scratch_bytes.restype = C.c_size_t
    L.hexvec_mlp_pack.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp]
    L.hexvec_mlp_actions.argtypes = [vp, vp, vp, vp, i32, vp, i32, i32, i32, vp, i32, i32, i32, i32, u64, u32, i32,
                                     vp, vp, vp, vp, vp, C.c_size_t, vp]
    L.hexvec_board_size.argtypes = [vp]
    L.hexvec_num_envs.argtypes = [vp]
    L.hexvec_flags.argtypes = [vp]
    L.hexvec_flags.restype = u32
    L.hexvec_last_error.restype = C.c_char_p
    L.hexvec_last_kernel.restype = C.c_char_p
    for name in SYMBOLS:
        if getattr(L, name).restype is C.c_int:
            pass
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise RuntimeError(f"libhexvec error {rc}: {lib().hexvec_last_error().decode()}")
