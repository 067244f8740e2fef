"""CPU-side checks: the C-ABI library loads and exports every symbol declared in
include/hexvec.h, fails loudly without a GPU, and the oracle's vectorised layer
is self-consistent.  No compute call of the product runs here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import hex_oracle as ho

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "hexvec.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(hexvec_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from hex_gym_env_amd import _lib
    so = _lib.build()
    L = C.CDLL(so)
    names = _declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(L, n), n
    assert sorted(_lib.SYMBOLS) == names


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import hex_gym_env_amd as hx
    with pytest.raises(RuntimeError):
        hx.HexVecEnv(5, 4)
    L = hx._lib.lib()
    h = C.c_void_p()
    rc = L.hexvec_create(C.byref(h), 5, 4, 0, C.c_uint64(0), 1, 0)
    assert rc < 0 and not h
    assert b"no CPU path" in L.hexvec_last_error() or rc == -2


def test_create_rejects_bad_arguments_before_touching_the_gpu():
    from hex_gym_env_amd import _lib
    L = _lib.lib()
    h = C.c_void_p()
    assert L.hexvec_create(C.byref(h), 20, 4, 0, C.c_uint64(0), 1, 0) == -1
    assert L.hexvec_create(C.byref(h), 15, 4, 0, C.c_uint64(0), 1 | 0x8, 0) == -1    # no auto reset: 2..14
    assert L.hexvec_create(C.byref(h), 15, 4, 0, C.c_uint64(0), 1 | 0x40, 0) == -1   # exact labels: 2..14
    assert L.hexvec_create(C.byref(h), 15, 4, 0, C.c_uint64(0), 1 | 0x80, 0) == -1   # rollout layout: 2..14
    assert L.hexvec_create(C.byref(h), 1, 4, 0, C.c_uint64(0), 1, 0) == -1
    assert L.hexvec_create(C.byref(h), 5, 0, 0, C.c_uint64(0), 1, 0) == -1
    assert L.hexvec_create(C.byref(h), 5, 4, 0, C.c_uint64(0), 0x2, 0) == -1   # enc A + WHITE agent
    assert L.hexvec_step(None, *([None] * 9)) == -1
    assert b"NULL" in L.hexvec_last_error()


@pytest.mark.parametrize("enc_flag", [0, ho.F_ENC_B])
def test_oracle_rollout_properties(enc_flag):
    n, E, T = 7, 128, 120
    o = ho.OracleVec(n, E, seed=3, flags=enc_flag | ho.F_TWO_SIDED)
    obs, legal, act, rew, done = o.rollout(T)
    empty = 2 if enc_flag == 0 else 0
    assert np.array_equal(legal, (obs == empty).astype(np.uint8))
    assert np.take_along_axis(legal, act[..., None].astype(np.int64), 2).all()
    assert np.array_equal(rew, done.astype(np.float32))
    st = o.episode_stats()
    assert st[2] == done.sum() and 35 < st[1] / st[2] < 50      # SURVEY: 42.2 plies at 7x7


def test_oracle_sharding_invariance_and_threads():
    n, E, T = 5, 64, 40
    full = ho.OracleVec(n, E, seed=11, flags=ho.F_ENC_B | ho.F_AGENT_RANDOM)
    half = ho.OracleVec(n, E // 2, seed=11, flags=ho.F_ENC_B | ho.F_AGENT_RANDOM, env_offset=E // 2)
    thr = ho.OracleVec(n, E, seed=11, flags=ho.F_ENC_B | ho.F_AGENT_RANDOM)
    thr.set_threads(4)
    assert np.array_equal(full.agent_colours()[E // 2:], half.agent_colours())
    for _ in range(T):
        a, b, c = full.step(), half.step(), thr.step()
        for x, y, z in zip(a, b, c):
            assert np.array_equal(x[E // 2:], y)
            assert np.array_equal(x, z)
    assert np.array_equal(full.episode_stats(), thr.episode_stats())


def test_oracle_rng_spec():
    """the counter RNG of include/hexvec.h, restated in numpy"""
    L = ho.lib()

    def mix(x):
        x = np.uint32(x)
        with np.errstate(over="ignore"):
            x ^= x >> np.uint32(16)
            x = np.uint32(x * np.uint32(0x7feb352d))
            x ^= x >> np.uint32(15)
            x = np.uint32(x * np.uint32(0x846ca68b))
            x ^= x >> np.uint32(16)
        return x

    for seed, env, ctr in [(0, 0, 0), (12345678901234567, 77, 5), (2 ** 64 - 1, 2 ** 31, 999)]:
        with np.errstate(over="ignore"):
            key = mix(np.uint32(env) * np.uint32(0x9E3779B9) + np.uint32(seed & 0xffffffff)) \
                ^ np.uint32(seed >> 32)
            draw = mix(key + np.uint32(ctr) * np.uint32(0x9E3779B9))
        assert L.hexo_rng_key(C.c_uint64(seed), env) == int(key)
        assert L.hexo_rng_draw(int(key), ctr) == int(draw)
    view = np.array([1, 0, 0, 1, 0, 1, 0, 0, 1], np.int8)
    p = view.ctypes.data_as(C.POINTER(C.c_int8))
    empties = np.flatnonzero(view == 0)
    for r in (0, 2 ** 31, 2 ** 32 - 1, 123456789):
        assert L.hexo_uniform_legal(p, 9, 0, r) == empties[(r * len(empties)) >> 32]


def test_minihex_alias_package_has_the_reference_module_names():
    import hex_gym_env_amd.minihex as m
    from hex_gym_env_amd.minihex.HexGame import HexEnv as EnvA, HexGame, player
    from hex_gym_env_amd.minihex.HexSingleGame import HexEnv as EnvB
    from hex_gym_env_amd.minihex.SelfplayWrapper import BaseRandomPolicy, OpponentPolicy, selfplay_wrapper
    assert (player.BLACK, player.WHITE, player.EMPTY) == (0, 1, 2)
    board = 2 * np.ones((3, 3))
    board[1, 1] = 0
    import random
    random.seed(1)
    assert m.random_policy(board) != 4                     # never the occupied cell
    assert callable(selfplay_wrapper) and BaseRandomPolicy().save_model("x") is None
    assert EnvA is not EnvB and HexGame is not None and OpponentPolicy is not None


def test_committed_profile_matches_the_drivers_command():
    """bench.py fills roofline.traffic / on_chip from profiles/<round>/bench_pmc_summary.json only if that profile was
    taken with the same board / envs / steps per launch: the committed one must be the driver's command"""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "bench.py")).read()
    rnd = src.split('PROFILE_ROUND = "')[1].split('"')[0]
    d = json.load(open(os.path.join(root, "profiles", rnd, "bench_pmc_summary.json")))
    assert d["config"] == {"board": 11, "envs": 65536, "steps_per_launch": 20}
    c = d["counters"]["rollout1"]
    for k in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_WAVES"):
        assert c[k] > 0, k
    assert d["kernel"].startswith("k_lpb")


# ---- bench.py --gpus N: the ranks are started by bench.py itself when no launcher did ----
def _bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_bench_gpus_n_without_enough_gpus_refuses_instead_of_reporting_one():
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode != 0
    assert "n_gpus" not in p.stdout and "needs 2 visible GPUs" in p.stderr


def test_bench_self_launch_builds_the_drivers_command_and_relays_rank0():
    import json
    import subprocess
    import types
    b = _bench()
    cmd = b.launcher_command(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], 29517)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29517"
    assert cmd[-7].endswith("bench.py") and cmd[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    seen = {}

    def fake_run(cmd, **kw):
        seen["cmd"], seen["env"] = cmd, kw["env"]
        line = json.dumps({"metric": "m", "value": 1.0, "n_gpus": seen["n"], "config": {"global_envs": 8 * 65536}})
        return subprocess.CompletedProcess(cmd, seen["rc"], stdout="noise from a rank\n" + line + "\n")

    args = types.SimpleNamespace(gpus=8, dry_ranks=False)
    seen.update(n=8, rc=0)
    assert b.self_launch(args, ["--gpus", "8", "--force-launcher"], run=fake_run, visible=8) == 0
    assert "--force-launcher" not in seen["cmd"] and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    seen.update(n=1, rc=0)          # ranks that report another size: refused
    assert b.self_launch(args, ["--gpus", "8"], run=fake_run, visible=8) != 0
    seen.update(n=8, rc=1)          # a failing launch keeps its exit code
    assert b.self_launch(args, ["--gpus", "8"], run=fake_run, visible=8) == 1
    assert b.self_launch(args, ["--gpus", "8"], run=fake_run, visible=4) == 2      # fewer GPUs than asked for


def test_bench_gpus_2_starts_two_real_ranks_and_prints_n_gpus_2():
    """the whole self-launch path with real processes: `bench.py --gpus 2 --dry-ranks` (no launcher environment)
    starts two ranks under torch.distributed.run as a child, the ranks run the bench's rank plumbing (process group,
    barrier, max over ranks, the stats all_gather) over gloo, and rank 0's line comes back with n_gpus == 2"""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-ranks", "--envs", "65536"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_envs"] == 2 * 65536 and d["dry"] is True
    assert d["max_over_ranks"] == 2.0 and d["stats_total"] == [2.0, 4.0, 6.0, 1.0]
    # under a launcher whose world size differs from --gpus the ranks refuse
    env2 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-ranks"],
                       capture_output=True, text=True, timeout=300, env=env2)
    assert p.returncode != 0 and "n_gpus" not in p.stdout
