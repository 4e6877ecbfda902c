"""Synthetic IQ streams of the measurement plan.

Per band: complex white Gaussian noise (sigma = 1e-3 per component) plus P on/off-keyed complex
exponentials of amplitude 0.1, each on an exact FFT bin centre and hard-keyed on frame boundaries
with the Morse text "cq de dl1abc dl1abc k" (dit = ceil(60 ms / tick) whole frames, random whole-frame
start offset).  Because every tone sits on a bin centre and a frame holds an integer number of its
periods, a frame's tone mix is the inverse FFT of a sparse spectrum — which is how it is generated.

Layout of the result: float32 [n_frames, 2*N], interleaved I,Q — the reference's []float32 frames
(dsp/fft.go:59-69).
"""
from __future__ import annotations

import math

import numpy as np

TEXT = "cq de dl1abc dl1abc k"
TONE_AMPLITUDE = 0.1
NOISE_SIGMA = 1e-3

_MORSE = {
    "a": ".-", "b": "-...", "c": "-.-.", "d": "-..", "e": ".", "f": "..-.", "g": "--.", "h": "....", "i": "..",
    "j": ".---", "k": "-.-", "l": ".-..", "m": "--", "n": "-.", "o": "---", "p": ".--.", "q": "--.-", "r": ".-.",
    "s": "...", "t": "-", "u": "..-", "v": "...-", "w": ".--", "x": "-..-", "y": "-.--", "z": "--..",
    "0": "-----", "1": ".----", "2": "..---", "3": "...--", "4": "....-", "5": ".....", "6": "-....", "7": "--...",
    "8": "---..", "9": "----.",
}


def keying_pattern(text: str, dit_frames: int) -> np.ndarray:
    """One pass of `text` as a 0/1 array, 1:3:1:3:7 timing, ending with a word gap."""
    out = []
    first = True
    for ch in text:
        if ch == " ":
            out += [0] * (7 * dit_frames)
            first = True
            continue
        if not first:
            out += [0] * (3 * dit_frames)
        for i, sym in enumerate(_MORSE[ch]):
            if i:
                out += [0] * dit_frames
            out += [1] * (dit_frames * (1 if sym == "." else 3))
        first = False
    out += [0] * (7 * dit_frames)
    return np.array(out, np.uint8)


def dit_frames(sample_rate: int, block_size: int) -> int:
    return max(2, math.ceil(0.060 / (block_size / sample_rate)))


def default_edge_width(block_size: int) -> int:
    return 70 * block_size // 512


def tone_bins(block_size: int, n_tones: int, edge_width: int | None = None,
              free_last_window: bool = False) -> np.ndarray:
    """Spectrum indices (after fftshift) spread uniformly over [edge+8, N-edge-8], >= 8 bins apart.

    free_last_window: keep the last of the reference's ten noise-floor windows (dsp/fft.go:216) free
    of tones.  The reference estimates the noise floor as the MINIMUM window mean; with hundreds of
    tones spread over all ten windows (BASELINE config 3: 256 tones at N=16384) every window holds
    keyed-down carriers, the "noise floor" lands ~60 dB too high and the listen threshold ends up
    ABOVE the carriers — nothing is detected, by the reference's own arithmetic (checked with the
    oracle).  Dense workloads therefore leave one window empty, as a real band would.
    """
    e = default_edge_width(block_size) if edge_width is None else edge_width
    lo, hi = e + 8, block_size - e - 8
    if free_last_window:
        hi = e + 9 * ((block_size - 2 * e) // 10) - 8
    if n_tones == 0:
        return np.zeros(0, np.int64)
    bins = np.unique(np.round(np.linspace(lo, hi, n_tones)).astype(np.int64))
    if len(bins) != n_tones or (n_tones > 1 and np.min(np.diff(bins)) < 8):
        raise ValueError("too many tones for this block size")
    return bins


def keying_matrix(n_frames: int, n_tones: int, sample_rate: int, block_size: int, rng: np.random.Generator,
                  text: str = TEXT) -> np.ndarray:
    """uint8 [n_frames, n_tones]: key state of every tone in every frame."""
    pat = keying_pattern(text, dit_frames(sample_rate, block_size))
    key = np.zeros((n_frames, n_tones), np.uint8)
    for p in range(n_tones):
        off = int(rng.integers(0, len(pat)))
        idx = (np.arange(n_frames) + off) % len(pat)
        key[:, p] = pat[idx]
    return key


def make_band(n_frames: int, sample_rate: int, block_size: int, n_tones: int, seed: int,
              edge_width: int | None = None, text: str = TEXT, noise_sigma: float = NOISE_SIGMA,
              amplitude: float = TONE_AMPLITUDE, free_last_window: bool = False):
    """Returns (iq float32 [n_frames, 2N], bins int64 [n_tones], key uint8 [n_frames, n_tones])."""
    rng = np.random.default_rng(seed)
    N = block_size
    bins = tone_bins(N, n_tones, edge_width, free_last_window)
    key = keying_matrix(n_frames, n_tones, sample_rate, N, rng, text)
    iq = np.empty((n_frames, 2 * N), np.float32)
    chunk = max(1, (1 << 22) // N)
    fft_bins = (bins + N // 2) % N  # spectrum index k = (i + N/2) % N  (dsp/fft.go:54-57)
    for f0 in range(0, n_frames, chunk):
        f1 = min(n_frames, f0 + chunk)
        spec = np.zeros((f1 - f0, N), np.complex128)
        if n_tones:
            spec[:, fft_bins] = amplitude * N * key[f0:f1].astype(np.float64)
        x = np.fft.ifft(spec, axis=1)
        x += noise_sigma * (rng.standard_normal((f1 - f0, N)) + 1j * rng.standard_normal((f1 - f0, N)))
        iq[f0:f1, 0::2] = x.real
        iq[f0:f1, 1::2] = x.imag
    return iq, bins, key


def make_band_torch(n_frames: int, sample_rate: int, block_size: int, n_tones: int, seed: int, device,
                    edge_width: int | None = None, text: str = TEXT, free_last_window: bool = False):
    """Same signal model generated directly in HBM with torch (benchmark input; not bit-identical to
    make_band, which is the one the parity tests share with the oracle)."""
    import torch

    rng = np.random.default_rng(seed)
    N = block_size
    bins = tone_bins(N, n_tones, edge_width, free_last_window)
    key = keying_matrix(n_frames, n_tones, sample_rate, N, rng, text)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty((n_frames, N, 2), dtype=torch.float32, device=device)
    fft_bins = torch.as_tensor((bins + N // 2) % N, device=device)
    chunk = max(1, (1 << 24) // N)
    for f0 in range(0, n_frames, chunk):
        f1 = min(n_frames, f0 + chunk)
        spec = torch.zeros((f1 - f0, N), dtype=torch.complex64, device=device)
        if n_tones:
            k = torch.as_tensor(key[f0:f1].astype(np.float32), device=device)
            spec[:, fft_bins] = (TONE_AMPLITUDE * N * k).to(torch.complex64)
        x = torch.fft.ifft(spec, dim=1)
        noise = torch.randn((f1 - f0, N, 2), generator=g, device=device, dtype=torch.float32) * NOISE_SIGMA
        out[f0:f1, :, 0] = x.real + noise[:, :, 0]
        out[f0:f1, :, 1] = x.imag + noise[:, :, 1]
    return out.reshape(n_frames, 2 * N), bins, key
