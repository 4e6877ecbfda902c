"""bench.py --gpus N starts its own ranks (spawn_ranks): when one of them dies the others - which would wait in the
rendezvous for ever, holding their GPUs - are stopped, the parent reports failure with every rank's output tail, and it
does so promptly.  No GPU needed: the test ranks die or hang before they import torch."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_a_dead_rank_takes_the_others_down():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(SDR_BENCH_TEST_DIE_RANK="1", SDR_BENCH_TEST_HANG_RANK="0")
    t0 = time.monotonic()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    took = time.monotonic() - t0
    assert p.returncode == 1, (p.returncode, p.stderr[-1500:])
    assert "ranks failed (rank, exit code): [(1, 3)]" in p.stderr and "rank 0 (exit" in p.stderr, p.stderr[-1500:]
    assert took < 60, f"the parent took {took:.0f} s to give up"
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")], "no result line from a failed world"
