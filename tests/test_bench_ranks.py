"""bench.py's N > 1 branch, rehearsed on ONE GPU: two fresh child processes (ranks 0 and 1 of a
world of 2) share the card, the collectives run over gloo on the CPU (a one-GPU box cannot host two
RCCL ranks).  Checks what the multi-GPU contract needs: one JSON line from rank 0 with n_gpus 2, whole-job
throughput over both ranks, each rank working on its own band, the broadcast configuration equal on both."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_bench_on_one_gpu(tmp_path):
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), SDR_DIST_BACKEND="gloo", SDR_FORCE_DEVICE="0",
                   SDR_BENCH_RANK_REPORT=str(tmp_path / f"rank{rank}.json"))
        # children are started before this process touches the GPU (it never does)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5",
                                       "--warmup", "2", "--frames", "256", "--settle-ms", "0", "--no-cpu-baseline"],
                                      env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for rank, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{se[-2000:]}"
    lines = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1, outs[0][0]
    assert not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")], "only rank 0 prints the line"
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["scaling"] == "weak" and res["steps"] == 5
    per_rank = res["config"]["samples_per_step_per_gpu"]
    assert res["value"] == pytest.approx(2 * per_rank * 5 / (res["ms_per_step"] * 5 * 1e-3) / 1e6, rel=1e-3)
    reports = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(2)]
    assert reports[0]["bands"] != reports[1]["bands"] and sorted(reports[0]["bands"] + reports[1]["bands"]) == [0, 1]
    assert reports[0]["shared_config"] == reports[1]["shared_config"]
    assert all(r["decoded_runes"] > 0 for r in reports)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE: bench.py starts its two ranks itself (before it
    touches the GPU), relays rank 0's single line and reports the world it really ran in."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(SDR_DIST_BACKEND="gloo", SDR_FORCE_DEVICE="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                        "--frames", "256", "--settle-ms", "0", "--no-cpu-baseline"], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 5
    d = res["config"]["distributed"]
    assert d["world_size"] == 2 and d["backend"] == "gloo" and d["launched_by"] == "bench.py itself"
    # every rank's own rate is in the line (value comes from the max over ranks: a straggler must be visible)
    pr = res["config"]["per_rank"]
    assert len(pr["ms_per_step"]) == 2 and len(pr["value"]) == 2 and pr["value_min"] <= pr["value_max"]
    assert max(pr["ms_per_step"]) <= res["ms_per_step"] * 1.001
    assert res["config"]["cpu_affinity_rank0"]["policy"]
    # a world that is not the one asked for is refused, never reported under another n_gpus
    env1 = dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    q = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline"], env=env1, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=600)
    assert q.returncode != 0 and "WORLD_SIZE=1" in (q.stderr + q.stdout)


def test_two_rank_bench_over_rccl():
    """The real thing where there are two GPUs: `python bench.py --gpus 2` over the nccl (= RCCL) backend, one rank per
    GPU - sharding.broadcast_config's collective travels over xGMI.  Skipped on a one-GPU box (the count is read in a
    child process: this one must not touch the GPU before it starts its ranks)."""
    n = int(subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True,
                           timeout=600).stdout.strip() or 0)
    if n < 2:
        pytest.skip(f"{n} GPU(s) visible: RCCL needs one rank per GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "SDR_DIST_BACKEND",
                                                              "SDR_FORCE_DEVICE")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                        "--frames", "512", "--settle-ms", "0", "--no-cpu-baseline"], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    res = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert res["n_gpus"] == 2 and res["config"]["distributed"]["backend"] == "nccl"
    assert len(res["config"]["per_rank"]["value"]) == 2
