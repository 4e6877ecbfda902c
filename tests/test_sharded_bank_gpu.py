"""sharding.ShardedBank on real banks: two ranks share ONE GPU (gloo for the collective: a one-GPU box cannot host two
RCCL ranks), each with its own band.  Between batches every rank calls the job-wide setter with a value of its own; rank
0's must reach both banks at the same batch boundary: both bands' peak_thr records change at the same frame and equal the
oracle's with the threshold set there (rx/receiver.go:208-211 applied between frames, :385)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle as orc
from sdrainer_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_collective_setter_reaches_both_banks_at_the_same_frame(tmp_path):
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "sharded_rank.py"), str(tmp_path / f"r{rank}.json")],
                                      env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for rank, p in enumerate(procs):
        so, se = p.communicate(timeout=600)
        assert p.returncode == 0, f"rank {rank}:\n{se[-2000:]}"
    res = [json.load(open(tmp_path / f"r{r}.json")) for r in range(2)]
    N, RATE, TONES, PER = 1024, 96000, 3, 150  # (as in tests/sharded_rank.py)
    edge = synth.default_edge_width(N)
    for r in res:
        assert r["cfg"]["peak_threshold"] == 21.5 and r["job_bands"] == [0.0, 1.0]
        iq, bins, _ = synth.make_band(3 * PER, RATE, N, TONES, seed=7000 + r["band"])
        ref = orc.Receiver(RATE, N, edge, 15.0, 1)
        for b in bins:
            ref.attach(int(b))
        for k, t in enumerate((15.0, 9.0, 21.5)):
            ref.set_peak_threshold(t)
            out = ref.process(iq[k * PER:(k + 1) * PER])
            want = out["frames"]["peak_thr"].view(np.uint32).tolist()
            assert r["peak_thr_bits"][k] == want, (r["rank"], k)
    assert res[0]["band"] == 0 and res[1]["band"] == 1
