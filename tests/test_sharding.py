"""Multi-process (world_size 2, gloo, CPU) coverage of the N>1 path: band partition, the broadcast of
the shared configuration from rank 0 and the gather of per-band records."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sdrainer_amd import sharding


def test_band_partition():
    for world in (1, 2, 4, 8):
        seen = []
        for r in range(world):
            mine = sharding.bands_of_rank(64, world, r)
            assert all(sharding.rank_of_band(b, world) == r for b in mine)
            assert len(mine) == 64 // world
            seen += mine
        assert sorted(seen) == list(range(64))
    assert sharding.bands_of_rank(8, 8, 3) == [3]


def test_config_pack_round_trip():
    c = sharding.SharedConfig(2_000_000, 16384, 2240, 12.5, 3, 256)
    assert sharding.SharedConfig.unpack(c.pack()) == c


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    # rank 0 holds the authoritative thresholds; rank 1 starts with stale ones
    cfg = sharding.SharedConfig(2_000_000, 16384, 2240, 15.0, 1, 256) if rank == 0 else sharding.SharedConfig()
    cfg = sharding.broadcast_config(cfg, dist, dev)
    bands = sharding.bands_of_rank(4, world, rank)
    local = np.stack([sharding.make_record(b, 2048, 2048 * cfg.block_size, 250 + b, 10 * b, b, 33.9, 60.9)
                      for b in bands])
    allr = sharding.gather_records(local, dist, dev)
    first = sharding.describe(cfg)
    # a setter call on rank 0 is re-broadcast
    if rank == 0:
        cfg.peak_threshold = 9.0
    cfg2 = sharding.broadcast_config(cfg, dist, dev)
    q.put((rank, first, bands, allr.tolist(), cfg2.peak_threshold))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_broadcast_and_gather_world2():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=100) for _ in range(world))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    want = sharding.describe(sharding.SharedConfig(2_000_000, 16384, 2240, 15.0, 1, 256))
    assert res[0][1] == want and res[1][1] == want
    assert res[0][2] == [0, 2] and res[1][2] == [1, 3]
    assert res[0][3] == res[1][3]
    assert [r[0] for r in res[0][3]] == [0.0, 1.0, 2.0, 3.0]
    assert [r[3] for r in res[0][3]] == [250.0, 251.0, 252.0, 253.0]
    assert res[0][4] == res[1][4] == 9.0


class _FakeBank:
    """Records what a rank's bank is told (the collective logic needs no GPU)."""

    def __init__(self, n_bands):
        self.n_bands = n_bands
        self.calls = []

    def set_peak_threshold(self, band, t):
        self.calls.append(("peak_threshold", band, t))

    def set_edge_width(self, e):
        self.calls.append(("edge_width", e))

    def set_signal_debounce(self, band, d):
        self.calls.append(("signal_debounce", band, d))


def _sharded_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    # rank 1 starts from a stale configuration and proposes its own values in every setter: rank 0's must win
    start = sharding.SharedConfig(2_000_000, 16384, 2240, 15.0, 1, 256) if rank == 0 else sharding.SharedConfig()
    bank = _FakeBank(n_bands=2)
    sb = sharding.ShardedBank(bank, start, dist, dev, n_bands_total=4)
    log = [list(bank.calls)]
    bank.calls.clear()
    sb.set_peak_threshold(9.0 if rank == 0 else 22.0)
    log.append(list(bank.calls))
    bank.calls.clear()
    sb.set_signal_debounce(3 if rank == 0 else 7)
    sb.set_edge_width(2000 if rank == 0 else 10)
    log.append(list(bank.calls))
    recs = np.stack([sharding.make_record(b, 100, 100 * 16384, b, 0, 0, 1.0, 2.0) for b in sb.bands])
    q.put((rank, sb.bands, log, sharding.describe(sb.cfg), sb.gather(recs)[:, 0].tolist(), sb.setter_calls))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_sharded_bank_setters_are_collective_world2():
    """ShardedBank: a setter called on every rank at the same point applies RANK 0's value to every rank's bank
    (rx/receiver.go:208-218 across processes), band by band, and only what changed is sent to the bank."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=100) for _ in range(world))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2] and res[1][1] == [1, 3]
    for rank, bands, log, cfg, gathered, calls in res:
        # construction: rank 0's configuration, every field, both local bands
        assert ("peak_threshold", 0, 15.0) in log[0] and ("peak_threshold", 1, 15.0) in log[0] and ("edge_width", 2240) in log[0]
        assert log[1] == [("peak_threshold", 0, 9.0), ("peak_threshold", 1, 9.0)]
        assert log[2] == [("signal_debounce", 0, 3), ("signal_debounce", 1, 3), ("edge_width", 2000)]
        assert cfg["peak_threshold"] == 9.0 and cfg["signal_debounce"] == 3 and cfg["edge_width"] == 2000 and cfg["block_size"] == 16384
        assert gathered == [0.0, 1.0, 2.0, 3.0] and calls == 3
