"""bench.py's stage lines and deadlines, as far as a box without a GPU can show them: a rank that
stalls costs a message naming the stage and a non-zero exit within the stage's deadline -- from the rank's own watchdog,
and from self_launch()'s backstop when the rank's interpreter is wedged -- and a launch whose WORLD_SIZE is not --gpus
refuses to print a line.  (The stall is KIFS_BENCH_STALL's sleep in the `import` stage, before anything touches a GPU;
the GPU half -- a rank asleep in the calibration or in the timed steps of a real two-rank run -- is in
tests/test_gpu_bench.py.)"""
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _bench(*args, env=None, timeout=120):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    t0 = time.time()
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True, timeout=timeout,
                       env=e, cwd=str(ROOT))
    return p, time.time() - t0


def test_a_stalled_stage_ends_the_run_within_its_deadline_and_is_named():
    # deadline of `import`: 420 s x 0.01 = 4.2 s
    p, took = _bench("--gpus", "2", env={"KIFS_BENCH_STALL": "import:*", "KIFS_BENCH_DEADLINE_SCALE": "0.01"})
    assert p.returncode != 0 and took < 40, (p.returncode, took)
    assert "stage: import | deadline 4 s" in p.stderr
    assert "STALLED in stage 'import'" in p.stderr and "giving up" in p.stderr
    assert "last stage of every rank" in p.stderr and '"0": "\'import\'' in p.stderr and '"1": "\'import\'' in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]  # no line for a run that did not happen


def test_the_parent_stops_ranks_whose_own_watchdog_is_silent():
    p, took = _bench("--gpus", "2", env={"KIFS_BENCH_STALL": "import:*", "KIFS_BENCH_DEADLINE_SCALE": "0.01",
                                          "KIFS_BENCH_NO_RANK_WATCHDOG": "1"})
    assert p.returncode != 0 and took < 40, (p.returncode, took)
    assert "STALLED" not in p.stderr
    assert "stalled in stage 'import'" in p.stderr and "last stage of every rank" in p.stderr


def test_one_stalled_rank_does_not_outlive_the_others_failure():
    # rank 1 sleeps; rank 0 goes on, finds no GPU in this container (or, on a GPU box, waits for rank 1 in the rendezvous
    # until the deadline) and fails: the parent stops rank 1 and says where everybody was
    p, took = _bench("--gpus", "2", env={"KIFS_BENCH_STALL": "import:1", "KIFS_BENCH_DEADLINE_SCALE": "0.05"}, timeout=200)
    assert p.returncode != 0 and took < 120
    assert "last stage of every rank" in p.stderr and '"1": "\'import\'' in p.stderr


def test_a_world_of_the_wrong_size_prints_no_line():
    p, _ = _bench("--gpus", "4", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                                     "MASTER_PORT": "29999"})
    assert p.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
