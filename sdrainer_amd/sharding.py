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


class ShardedBank:
    """One rank's bank plus the job's shared configuration (rx/receiver.go:166-172,208-218: the reference's setters are
    closures marshalled to the run goroutine and applied between frames; here a job is N processes, one per GPU, each with
    the bands b = rank (mod world), and a setter is a COLLECTIVE: every rank calls it at the same batch boundary - the
    ranks run the same program - rank 0's argument wins, it is broadcast (RCCL over xGMI with backend "nccl", gloo in the
    CPU tests) and every rank applies it to its own bank through sdr_set_*, which takes effect at the next batch, never
    inside one.  So every band of the job changes its threshold at the same frame.

    `bank` is a sdrainer_amd.capi.Bank (or anything with set_peak_threshold(band, t), set_edge_width(e),
    set_signal_debounce(band, d) and n_bands); `dist` torch.distributed or None."""

    def __init__(self, bank, cfg: SharedConfig, dist, device, n_bands_total: int | None = None):
        self.bank, self.dist, self.device = bank, dist, device
        self.rank = dist.get_rank() if self._distributed() else 0
        self.world = dist.get_world_size() if self._distributed() else 1
        self.n_bands_total = n_bands_total if n_bands_total is not None else self.world * bank.n_bands
        self.bands = bands_of_rank(self.n_bands_total, self.world, self.rank)
        self.cfg = broadcast_config(cfg, dist, device)
        self.setter_calls = 0
        self._apply(self.cfg, None)  # (the bank gets every one of rank 0's values)

    def _distributed(self) -> bool:
        return self.dist is not None and self.dist.is_initialized() and self.dist.get_world_size() > 1

    def _apply(self, new: SharedConfig, old: SharedConfig | None):
        for local in range(self.bank.n_bands):
            if old is None or new.peak_threshold != old.peak_threshold:
                self.bank.set_peak_threshold(local, float(new.peak_threshold))
            if old is None or new.signal_debounce != old.signal_debounce:
                self.bank.set_signal_debounce(local, int(new.signal_debounce))
        if old is None or new.edge_width != old.edge_width:
            self.bank.set_edge_width(int(new.edge_width))

    def _collective_set(self, **changes):
        """Rank 0's `changes` reach every rank's bank; what the other ranks passed is ignored."""
        proposal = SharedConfig(**{**asdict(self.cfg), **changes})
        new = broadcast_config(proposal, self.dist, self.device)
        old, self.cfg = self.cfg, new
        self._apply(new, old)
        self.setter_calls += 1
        return new

    def set_peak_threshold(self, threshold: float):  # rx/receiver.go:208-211
        return self._collective_set(peak_threshold=float(threshold))

    def set_edge_width(self, edge_width: int):  # rx/receiver.go:217-218
        return self._collective_set(edge_width=int(edge_width))

    def set_signal_debounce(self, debounce: int):  # rx/receiver.go:212-216
        return self._collective_set(signal_debounce=int(debounce))

    def gather(self, local_records: np.ndarray) -> np.ndarray:
        """Every rank's fixed-size per-band records on every rank, by band id."""
        return gather_records(local_records, self.dist, self.device)
