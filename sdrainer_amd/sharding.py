"""Band sharding for one-process-per-GPU runs.

Every band is an independent receiver (rx/receiver.go:64-91 owns all of its state), so the job shards
by band: band b -> rank b mod world.  There is no exchange step inside the DSP, hence no data-path
collective.  The only shared state is the small configuration / threshold struct the reference's
setters change (peak threshold, edge width, debounce, pool size): rank 0 owns it and broadcasts it over
RCCL (xGMI) at start and whenever a setter is called.  Optionally each rank's fixed-size per-band
result records are gathered on every rank.  Works with backend "nccl" (= RCCL on ROCm) on GPUs and
with "gloo" on CPU (tests).
"""
from __future__ import annotations

from dataclasses import asdict, dataclass, fields

import numpy as np

RECORD_WIDTH = 8  # float64 words per band result record


@dataclass
class SharedConfig:
    sample_rate: int = 48000
    block_size: int = 512
    edge_width: int = 70            # rx/receiver.go:25
    peak_threshold: float = 15.0    # rx/receiver.go:24
    signal_debounce: int = 1        # cw/spectral.go:14
    max_listeners: int = 30         # rx/receiver.go:26

    def pack(self) -> np.ndarray:
        return np.array([float(getattr(self, f.name)) for f in fields(self)], np.float64)

    @classmethod
    def unpack(cls, a) -> "SharedConfig":
        vals = {}
        for f, v in zip(fields(cls), a):
            vals[f.name] = float(v) if f.type in (float, "float") else int(round(float(v)))
        return cls(**vals)


def bands_of_rank(n_bands: int, world: int, rank: int) -> list[int]:
    """Band b lives on rank b mod world."""
    return [b for b in range(n_bands) if b % world == rank]


def rank_of_band(band: int, world: int) -> int:
    return band % world


def broadcast_config(cfg: SharedConfig, dist, device) -> SharedConfig:
    """Rank 0's configuration wins on every rank.  `dist` is torch.distributed (initialised) or None."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return cfg
    import torch

    t = torch.from_numpy(cfg.pack()).to(device)
    dist.broadcast(t, src=0)
    return SharedConfig.unpack(t.cpu().numpy())


def make_record(band: int, frames: int, samples: int, peaks: int, edges: int, runes: int, noise_floor: float,
                listen_thr: float) -> np.ndarray:
    return np.array([band, frames, samples, peaks, edges, runes, noise_floor, listen_thr], np.float64)


def gather_records(local: np.ndarray, dist, device) -> np.ndarray:
    """local: float64 [bands_on_this_rank, RECORD_WIDTH] -> all ranks' records, sorted by band id.

    Every rank must hold the same number of bands (pad with band id -1 otherwise)."""
    local = np.ascontiguousarray(local, np.float64).reshape(-1, RECORD_WIDTH)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        out = local
    else:
        import torch

        t = torch.from_numpy(local).to(device)
        parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, t)
        out = torch.cat(parts).cpu().numpy()
    out = out[out[:, 0] >= 0]
    return out[np.argsort(out[:, 0], kind="stable")]


def describe(cfg: SharedConfig) -> dict:
    return asdict(cfg)
