"""Every back-end switch that ships must still produce the oracle's factors: each setting runs in its own process (the
switches are read once) on a mid-size matrix at nb = 256 where the dense paths engage.  Tolerance 1e-12 on the factors
relative to their largest entry, residual and the reference's factor check (src/pangulu_numeric.c:1082-1341) below 1e-12."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SWITCHES = [
    {},  # defaults
    {"PANGULU_HIP_TRSM_DIRECT": "0"},          # LDS-staged dense solves
    {"PANGULU_HIP_TRSM_RING": "0"},            # round 4's barrier-free dense solves for TSTRF too (default: factor tiles requested ahead through LDS)
    {"PANGULU_HIP_TRSM_RING": "0", "_matrix": "fem27"},
    {"PANGULU_HIP_GETRF_PIPE": "0"},           # round 4's tiled GETRF (trailing tiles through L2) instead of the register-resident one
    {"PANGULU_HIP_GETRF_PIPE": "0", "_matrix": "fem27"},
    {"PANGULU_HIP_RECORDS_STREAM": "0"},       # sparsify jobs of finished blocks on the main stream
    {"PANGULU_HIP_OCCUPANCY_SUMMARIES": "0"},  # no host-side pattern summaries: maps read behind the mirrors, full work lists
    {"PANGULU_HIP_RESERVED_CUS": "8"},         # CU-masked bulk streams
    {"PANGULU_HIP_LAUNCH_CHUNK": "64"},        # launches cut into chunks of 64 tasks
    # launches of 8 tasks: a destination's queue is cut between launches, and on the background stream the mirror jobs of the later
    # launch must wait for the kernels of the earlier one (this case gave factors 2.8e-4 off until the end of round 4)
    {"PANGULU_HIP_LAUNCH_CHUNK": "8", "_matrix": "fem27"},
    {"PANGULU_HIP_LAUNCH_CHUNK": "8", "PG_TEST_HIP_OPTIONS": "2=100", "_matrix": "fem27"},  # ... with more updates on the sparse records
    {"PANGULU_HIP_LAUNCH_CHUNK": "8", "PG_TEST_HIP_OPTIONS": "2=100"},
    {"PANGULU_HIP_FRONT_FORK": "0", "_matrix": "fem27"},  # dense-front and general update launches on one stream (default since round 6: two)
    {"PANGULU_AMD_ASYNC_LAUNCH": "0"},         # platform calls on the scheduler thread
    {"PANGULU_AMD_LOOKAHEAD_MAX_GETRF": "0"},  # lazy updates: queues accumulate until the destination's own panel task
    {"PANGULU_AMD_LOOKAHEAD_MAX_GETRF": "0", "PANGULU_HIP_LAUNCH_CHUNK": "64"},
    {"PANGULU_AMD_LOOKAHEAD_MAX_GETRF": "0", "PANGULU_HIP_LAUNCH_CHUNK": "64", "_matrix": "fem27"},  # (long queues cut by launch chunks)
    {"PANGULU_AMD_LOOKAHEAD_MAX_GETRF": "0", "PANGULU_HIP_SMALL_LAUNCH_TASKS": "0", "_matrix": "fem27"},
    # round 5: look-ahead calls leave shallow update queues alone -- by default only from 8 192 queued updates on, forced here
    {"PANGULU_AMD_LOOKAHEAD_DEFER_FROM": "0"},
    {"PANGULU_AMD_LOOKAHEAD_DEFER_FROM": "0", "_matrix": "fem27"},
    {"PANGULU_AMD_LOOKAHEAD_DEFER_FROM": "0", "PANGULU_AMD_LOOKAHEAD_MIN_QUEUE": "8", "_matrix": "fem27"},
    {"PANGULU_AMD_LOOKAHEAD_DEFER_FROM": "0", "PANGULU_AMD_LOOKAHEAD_MIN_QUEUE": "1000", "PANGULU_HIP_LAUNCH_CHUNK": "8"},
    {"PANGULU_AMD_LOOKAHEAD_DEFER_FROM": "0", "PANGULU_AMD_LOOKAHEAD_MIN_TASKS": "512", "_matrix": "fem27"},
    {"PANGULU_AMD_LOOKAHEAD_MIN_QUEUE": "1"},  # round 4's behaviour
    {"PANGULU_AMD_PANEL_FIRST": "0", "PG_TEST_HIP_OPTIONS": "14=0"},  # round 2's look-ahead: no background stream
    {"PANGULU_AMD_PANEL_FIRST": "0"},
    {"PANGULU_AMD_REPLAY": "0"},               # the scheduler in the loop (no static schedule)
    {"PANGULU_AMD_RECORD_AT_INIT": "0"},       # schedule recorded by the first gstrf instead of a dry run at init
    {"PANGULU_AMD_FORCE_MULTI_LOOP": "1"},     # one rank through the multi-rank scheduler loop (launcher thread + markers)
    {"PG_TEST_HIP_OPTIONS": "15=0,16=0"},      # round 2's MFMA update kernel
    {"PG_TEST_HIP_OPTIONS": "15=3"},           # dense-front kernel with three LDS stages
    {"PANGULU_AMD_BIND_NUMA": "0"},
    {"HSA_ENABLE_INTERRUPT": "0"},             # what bench.py sets
    # the configuration bench.py TIMES: structural flop counting of the MFMA path off (the kernel gets a null product
    # counter), no per-launch events
    {"PG_TEST_HIP_OPTIONS": "6=0,3=0", "HSA_ENABLE_INTERRUPT": "0"},
    {"PG_TEST_HIP_OPTIONS": "6=0,3=0,10=0,13=0"},  # ... and bench.py's profile-pass stream layout: everything on one stream
    {"PG_TEST_HIP_OPTIONS": "3=1,10=0,13=0"},      # the profile pass itself
    # GETRF -> dense-solve chase (off by default): a level's factorisations and its dense solves in one launch
    {"PANGULU_HIP_CHASE": "1", "PANGULU_HIP_CHASE_MAX_GETRF": "4"},
    {"PANGULU_HIP_CHASE": "1", "PANGULU_HIP_CHASE_MAX_GETRF": "256", "_matrix": "fem27"},
    {"PANGULU_AMD_SEPARATOR_ORDER": "natural", "_matrix": "fem27"},  # separators in the mesh's numbering (default: k-d order)
    {"PANGULU_HIP_HEAVY_FIRST": "0", "_matrix": "fem27"},              # update work items in the scheduler's order
    {"PANGULU_HIP_SOLVE_CHUNKED": "0"},        # rounds 2-3's triangular-solve kernels (column by column from HBM)
    {"PANGULU_HIP_EARLY_DENSIFY": "1"},        # first-touch densify jobs in a prologue on their own stream (measured: no gain; off)
    {"PANGULU_HIP_EARLY_DENSIFY": "1", "_matrix": "fem27"},
    {"PANGULU_AMD_ND_DIAGONALS": "0", "PANGULU_AMD_ND_POLISH": "0", "_matrix": "fem27"},  # rounds 1-3's geometric cuts: axes only, no FM
    # the documented debug switch: phase stamps of the GETRF kernels.  ADVICE r5: the pipe kernel's slots lay 22 words
    # past the 16-word counter allocation -- a device write out of bounds that this case would have caught as a fault or as broken factors
    {"PANGULU_HIP_DEBUG_GETRF": "1"},
    {"PANGULU_HIP_DEBUG_GETRF": "1", "PANGULU_HIP_GETRF_PIPE": "0", "_matrix": "fem27"},
]


@pytest.mark.parametrize("env", SWITCHES, ids=["+".join("%s=%s" % kv for kv in e.items()) or "defaults" for e in SWITCHES])
def test_backend_switch_keeps_parity(env):
    e = dict(os.environ)
    env = dict(env)
    which = env.pop("_matrix", "shell")
    e.update(env)
    e["PYTHONPATH"] = ROOT + os.pathsep + e.get("PYTHONPATH", "")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "env_switch_worker.py"), which], env=e, cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert r["dL"] <= 1e-12 and r["dU"] <= 1e-12, r
    assert r["residual"] <= 1e-12 and r["lu_check"] <= 1e-12, r
    assert r["factor_check_device"] <= 1e-12, r  # pangulu_amd_factor_check on the device-resident records
    if "6=0" not in env.get("PG_TEST_HIP_OPTIONS", ""):
        assert r["flop_counted"] == r["flop"], r
    assert r["dense_updates"] > 0 and r["dense_solves"] > 0 and r["getrf_launches"] > 0, r
    if env.get("PANGULU_HIP_CHASE") == "1":
        assert r["chase_launches"] > 0 and r["chase_solves"] > 0, r


# ---------------------------------------------------------------------------------------------------------------------
# A seeded random sweep next to the hand-kept list above (the list had 37 combinations and the bug of round 4 -- mirror
# jobs of a cut background launch -- sat in a combination nobody had listed).  Every draw switches each option away from its default
# with probability 1/4, on one of the two matrices, with launches cut into chunks of 8 / 64 tasks or not at all.  The seed is in the
# test ids; PG_SWEEP_SEED / PG_SWEEP_DRAWS choose another sweep (e.g. a longer one before a release).
# ---------------------------------------------------------------------------------------------------------------------
SWEEP_SPACE = [
    # (environment variable or back-end option number, non-default values)
    ("PANGULU_HIP_LAUNCH_CHUNK", ["8", "64"]),
    ("PANGULU_HIP_TRSM_DIRECT", ["0"]),
    ("PANGULU_HIP_RECORDS_STREAM", ["0"]),
    ("PANGULU_HIP_OCCUPANCY_SUMMARIES", ["0"]),
    ("PANGULU_HIP_FRONT_FORK", ["0"]),
    ("PANGULU_HIP_HEAVY_FIRST", ["0", "1"]),
    ("PANGULU_HIP_EARLY_DENSIFY", ["1"]),
    ("PANGULU_HIP_SOLVE_CHUNKED", ["0"]),
    ("PANGULU_HIP_SMALL_LAUNCH_TASKS", ["0"]),
    ("PANGULU_HIP_CHASE", ["1"]),
    ("PANGULU_AMD_ASYNC_LAUNCH", ["0"]),
    ("PANGULU_AMD_LOOKAHEAD_MAX_GETRF", ["0", "4"]),
    ("PANGULU_AMD_PANEL_FIRST", ["0"]),
    ("PANGULU_AMD_REPLAY", ["0"]),
    ("PANGULU_AMD_RECORD_AT_INIT", ["0"]),
    ("PANGULU_AMD_FORCE_MULTI_LOOP", ["1"]),
    (2, ["10", "100"]),      # dense-update threshold (per mille): more updates on the sparse records
    (8, ["1", "16"]),        # queue chunks
    (9, ["50"]),             # dense-solve threshold
    (10, ["0"]),             # MFMA kernel beside the LDS kernel on a side stream
    (13, ["0"]),             # records stream
    (14, ["0"]),             # background updates
    (15, ["0", "3"]),        # dense-front kernel: off / three stages
    (16, ["0"]),             # general update kernel: round 2's (the others of rounds 3-5 are in tools/experiments/)
    ("PANGULU_HIP_GETRF_PIPE", ["0"]),  # (appended: the draws before this entry keep their settings for the options above)
    ("PANGULU_AMD_LOOKAHEAD_DEFER_FROM", ["0"]),
    ("PANGULU_AMD_LOOKAHEAD_MIN_QUEUE", ["1", "2", "8", "1000"]),
    ("PANGULU_HIP_TRSM_RING", ["0"]),
]


def sweep_draws():
    import random

    # Round 6: 24 draws per run instead of 60 (the GPU suite had grown to 706 s of the driver's 1200 s step limit), and the seed ROTATES
    # with the calendar day, so that successive runs of the suite cover different corners instead of repeating one sweep; the seed is in
    # every test id, PG_SWEEP_SEED=<that seed> reproduces a failure, PG_SWEEP_DRAWS=200 is the long sweep before a release.
    import datetime

    seed = int(os.environ.get("PG_SWEEP_SEED", str(20261003 + datetime.date.today().toordinal() % 7)))
    ndraw = int(os.environ.get("PG_SWEEP_DRAWS", "24"))
    rng = random.Random(seed)
    draws = []
    for k in range(ndraw):
        env, opts = {"_matrix": rng.choice(["shell", "fem27"])}, []
        for key, values in SWEEP_SPACE:
            if rng.random() < 0.25:
                val = rng.choice(values)
                if isinstance(key, int):
                    opts.append("%d=%s" % (key, val))
                else:
                    env[key] = val
        if opts:
            env["PG_TEST_HIP_OPTIONS"] = ",".join(opts)
        draws.append(pytest.param(env, id="seed%d-draw%02d" % (seed, k)))
    return draws


@pytest.mark.parametrize("env", sweep_draws())
def test_random_switch_sweep_keeps_parity(env, tmp_path_factory):
    e = dict(os.environ)
    env = dict(env)
    which = env.pop("_matrix")
    e.update(env)
    e["PYTHONPATH"] = ROOT + os.pathsep + e.get("PYTHONPATH", "")
    # (the oracle's factors once per matrix and session, not once per draw)
    e["PG_TEST_REF_CACHE"] = os.path.join(str(tmp_path_factory.getbasetemp()), "sweep_ref_%s.npz" % which)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "env_switch_worker.py"), which], env=e, cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, "%r\n%s" % (env, out.stderr[-2000:])
    r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert r["dL"] <= 1e-12 and r["dU"] <= 1e-12, (env, r)
    assert r["residual"] <= 1e-12 and r["lu_check"] <= 1e-12 and r["factor_check_device"] <= 1e-12, (env, r)
    assert r["getrf_launches"] > 0, (env, r)
