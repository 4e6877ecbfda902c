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
