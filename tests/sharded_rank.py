"""One rank of tests/test_sharded_bank_gpu.py (a script, not a test): a real bank on the GPU behind sharding.ShardedBank,
a collective setter between two batches, the batch's frame records written out for the parent to compare with the oracle.
usage: sharded_rank.py <out.json>      (RANK / WORLD_SIZE / MASTER_* in the environment, gloo)"""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sdrainer_amd import capi, sharding, synth  # noqa: E402

N, RATE, TONES, PER = 1024, 96000, 3, 150


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    edge = synth.default_edge_width(N)
    start = sharding.SharedConfig(RATE, N, edge, 15.0, 1, TONES) if rank == 0 else sharding.SharedConfig(RATE, N, edge, 40.0, 1, TONES)
    bank = capi.Bank(RATE, N, edge_width=edge, max_batch_frames=PER, max_listeners=TONES, max_peaks=64)
    sb = sharding.ShardedBank(bank, start, dist, dev, n_bands_total=world)
    band = sb.bands[0]
    iq, bins, _ = synth.make_band(3 * PER, RATE, N, TONES, seed=7000 + band)
    for b in bins:
        bank.attach(0, int(b))
    thr = []
    for k in range(3):
        if k == 1:  # between batch 0 and batch 1: every rank calls, rank 0's 9.0 wins over rank 1's 33.0
            sb.set_peak_threshold(9.0 if rank == 0 else 33.0)
        if k == 2:
            sb.set_peak_threshold(21.5 if rank == 0 else 1.0)
        assert bank.process_host(iq[k * PER:(k + 1) * PER]) == PER
        rec = bank.read_frame_records(0)
        thr.append(rec["peak_thr"].view(np.uint32).tolist())
    records = sb.gather(np.stack([sharding.make_record(band, 3 * PER, 3 * PER * N, 0, 0, 0, 0.0, 0.0)]))
    json.dump({"rank": rank, "band": band, "peak_thr_bits": thr, "cfg": sharding.describe(sb.cfg), "job_bands": records[:, 0].tolist()},
              open(sys.argv[1], "w"))
    bank.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
