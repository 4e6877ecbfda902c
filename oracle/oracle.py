"""ctypes wrapper around oracle/liborc.so (the CPU restatement of the reference).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (sdrainer_amd) never imports
this module.  See oracle/sdr_oracle.c for the pinning status of each function.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liborc.so")


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "sdr_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liborc.so"])
    return _LIB_PATH


class Peak(C.Structure):
    """dsp.Peak[float32,int] (dsp/fft.go:179-188) with fixed-width fields."""

    _fields_ = [
        ("from_", C.c_int32),
        ("to", C.c_int32),
        ("from_frequency", C.c_int64),
        ("to_frequency", C.c_int64),
        ("signal_frequency", C.c_int64),
        ("signal_value", C.c_float),
        ("signal_bin", C.c_int32),
    ]

    def astuple(self):
        return (self.from_, self.to, self.from_frequency, self.to_frequency, self.signal_frequency,
                float(np.float32(self.signal_value)), self.signal_bin)


class FrameRec(C.Structure):
    _fields_ = [
        ("min_mean", C.c_float),
        ("variance", C.c_double),
        ("dev_in", C.c_float),
        ("nf_in", C.c_float),
        ("noise_dev", C.c_float),
        ("noise_floor", C.c_float),
        ("peak_thr", C.c_float),
        ("listen_thr", C.c_float),
    ]


FRAME_REC_DTYPE = np.dtype(
    {
        "names": ["min_mean", "variance", "dev_in", "nf_in", "noise_dev", "noise_floor", "peak_thr", "listen_thr"],
        "formats": ["<f4", "<f8", "<f4", "<f4", "<f4", "<f4", "<f4", "<f4"],
        "offsets": [FrameRec.min_mean.offset, FrameRec.variance.offset, FrameRec.dev_in.offset, FrameRec.nf_in.offset,
                    FrameRec.noise_dev.offset, FrameRec.noise_floor.offset, FrameRec.peak_thr.offset,
                    FrameRec.listen_thr.offset],
        "itemsize": C.sizeof(FrameRec),
    }
)

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    dp, fp, u8p, u32p, ip = (C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_uint8),
                             C.POINTER(C.c_uint32), C.POINTER(C.c_int))

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)

    for n in ("orc_go_log", "orc_go_log2", "orc_go_log10"):
        sig(n, C.c_double, C.c_double)
    sig("orc_go_sincos", None, C.c_double, dp, dp)
    sig("orc_radix2_factors", None, C.c_int, dp, dp)
    sig("orc_fft_radix2", None, C.c_int, dp, dp, dp, dp)
    sig("orc_bin_to_spectrum_index", C.c_int, C.c_int, C.c_int)
    sig("orc_psd_value_in_db", C.c_float, C.c_float, C.c_int)
    sig("orc_iq_to_spectrum_and_psd", None, C.c_int, fp, fp, fp)
    sig("orc_iq_fft", None, C.c_int, fp, dp, dp)
    sig("orc_find_noise_floor", None, fp, C.c_int, C.c_int, fp, dp)
    sig("orc_bin_to_frequency", C.c_int64, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_double)
    sig("orc_frequency_to_bin", C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64)
    sig("orc_peak_center_correction", C.c_double, C.c_int, fp, C.c_int)
    sig("orc_find_peaks", C.c_int, fp, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int64, C.POINTER(Peak), C.c_int)
    sig("orc_decode_iq_message", C.c_long, C.c_char_p, C.c_size_t, fp)
    sig("orc_rolling_mean_new", C.c_void_p, C.c_int)
    sig("orc_rolling_mean_put", C.c_float, C.c_void_p, C.c_float)
    sig("orc_rolling_mean_free", None, C.c_void_p)
    sig("orc_debouncer_new", C.c_void_p, C.c_int)
    sig("orc_debouncer_debounce", C.c_int, C.c_void_p, C.c_int)
    sig("orc_debouncer_free", None, C.c_void_p)
    sig("orc_goertzel_blocksize", C.c_int, C.c_double, C.c_int, C.c_double)
    sig("orc_morse_count", C.c_int)
    sig("orc_morse_rune", C.c_uint32, C.c_int)
    sig("orc_morse_code", C.c_char_p, C.c_int)
    sig("orc_decoder_new", C.c_void_p, C.c_int, C.c_int)
    sig("orc_decoder_free", None, C.c_void_p)
    for n in ("orc_decoder_reset", "orc_decoder_clear", "orc_decoder_stop", "orc_decoder_out_reset"):
        sig(n, None, C.c_void_p)
    sig("orc_decoder_preset_wpm", None, C.c_void_p, C.c_int)
    sig("orc_decoder_tick", None, C.c_void_p, C.c_int)
    sig("orc_decoder_ticks", None, C.c_void_p, u8p, C.c_int)
    sig("orc_decoder_wpm", C.c_double, C.c_void_p)
    sig("orc_decoder_out_len", C.c_int, C.c_void_p)
    sig("orc_decoder_out", u32p, C.c_void_p)
    sig("orc_decoder_state", None, C.c_void_p, dp)
    sig("orc_generate_stream", C.c_int, C.c_int, C.c_int, C.c_int, ip, u32p, C.c_int, u8p, C.c_int)
    sig("orc_audio_new", C.c_void_p, C.c_double, C.c_int)
    sig("orc_audio_free", None, C.c_void_p)
    sig("orc_audio_blocksize", C.c_int, C.c_void_p)
    sig("orc_audio_coeff", C.c_double, C.c_void_p)
    sig("orc_audio_set_scale", None, C.c_void_p, C.c_double)
    sig("orc_audio_set_debounce", None, C.c_void_p, C.c_int)
    sig("orc_audio_set_magnitude_threshold", None, C.c_void_p, C.c_double)
    sig("orc_audio_write", C.c_int, C.c_void_p, fp, C.c_int, dp, u8p, u8p, C.c_int)
    sig("orc_audio_close", None, C.c_void_p)
    sig("orc_audio_out_len", C.c_int, C.c_void_p)
    sig("orc_audio_out", u32p, C.c_void_p)
    sig("orc_peaks_table_new", C.c_void_p, C.c_int)
    sig("orc_peaks_table_free", None, C.c_void_p)
    sig("orc_peaks_table_set_now", None, C.c_void_p, C.c_double)
    sig("orc_peaks_table_seed", None, C.c_void_p, C.c_uint64)
    sig("orc_peaks_table_put", C.c_int, C.c_void_p, C.c_int, C.c_int)
    sig("orc_peaks_table_force_put", C.c_int, C.c_void_p, C.c_int, C.c_int)
    sig("orc_peaks_table_place", C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int)
    sig("orc_peaks_table_at", C.c_int, C.c_void_p, C.c_int)
    sig("orc_peaks_table_state", C.c_int, C.c_void_p, C.c_int)
    sig("orc_peaks_table_cleanup", None, C.c_void_p)
    sig("orc_peaks_table_activate", None, C.c_void_p, C.c_int, C.c_int)
    sig("orc_peaks_table_deactivate", None, C.c_void_p, C.c_int, C.c_int)
    sig("orc_peaks_table_find_next", C.c_int, C.c_void_p)
    sig("orc_peaks_table_entry", None, C.c_void_p, C.c_int, ip, ip)
    sig("orc_receiver_new", C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int64)
    sig("orc_receiver_free", None, C.c_void_p)
    sig("orc_receiver_set_peak_threshold", None, C.c_void_p, C.c_float)
    sig("orc_receiver_set_edge_width", None, C.c_void_p, C.c_int)
    sig("orc_receiver_set_find_peaks", None, C.c_void_p, C.c_int)
    sig("orc_receiver_attach", C.c_int, C.c_void_p, C.c_int)
    sig("orc_receiver_detach", None, C.c_void_p, C.c_int)
    sig("orc_receiver_text_len", C.c_int, C.c_void_p, C.c_int)
    sig("orc_receiver_text", u32p, C.c_void_p, C.c_int)
    sig("orc_receiver_decoder_state", None, C.c_void_p, C.c_int, dp)
    sig("orc_receiver_cumulation", fp, C.c_void_p)
    sig("orc_receiver_cumulation_count", C.c_int, C.c_void_p)
    sig("orc_receiver_process", C.c_int, C.c_void_p, fp, C.c_int, C.POINTER(FrameRec), fp, fp, fp, u8p, u8p,
        C.POINTER(Peak), ip, ip, C.c_int, C.c_int, fp)
    sig("orc_receiver_run_baseline", C.c_int, C.c_void_p, fp, C.c_int)
    _lib = L
    return L


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct)) if a is not None else None


def runes_to_str(ptr, n) -> str:
    return "".join(chr(ptr[i]) for i in range(n))


# --- thin functional wrappers --------------------------------------------------------------

def go_log10(x: float) -> float:
    return lib().orc_go_log10(x)


def go_sincos(x: float):
    s, c = C.c_double(), C.c_double()
    lib().orc_go_sincos(x, C.byref(s), C.byref(c))
    return s.value, c.value


def radix2_factors(n: int) -> np.ndarray:
    re, im = np.empty(n), np.empty(n)
    lib().orc_radix2_factors(n, _p(re, C.c_double), _p(im, C.c_double))
    return re + 1j * im


def iq_fft(iq: np.ndarray) -> np.ndarray:
    iq = np.ascontiguousarray(iq, dtype=np.float32)
    n = iq.size // 2
    re, im = np.empty(n), np.empty(n)
    lib().orc_iq_fft(n, _p(iq, C.c_float), _p(re, C.c_double), _p(im, C.c_double))
    return re + 1j * im


def iq_to_spectrum_and_psd(iq: np.ndarray):
    iq = np.ascontiguousarray(iq, dtype=np.float32)
    n = iq.size // 2
    sp, psd = np.empty(n, np.float32), np.empty(n, np.float32)
    lib().orc_iq_to_spectrum_and_psd(n, _p(iq, C.c_float), _p(sp, C.c_float), _p(psd, C.c_float))
    return sp, psd


def find_noise_floor(psd: np.ndarray, edge: int):
    psd = np.ascontiguousarray(psd, dtype=np.float32)
    m, v = C.c_float(), C.c_double()
    lib().orc_find_noise_floor(_p(psd, C.c_float), psd.size, edge, C.byref(m), C.byref(v))
    return np.float32(m.value), v.value


def find_peaks(cum: np.ndarray, threshold: float, sample_rate: int, center: int = 0, cumulation_size: int = 100,
               max_peaks: int = 8192):
    cum = np.ascontiguousarray(cum, dtype=np.float32)
    arr = (Peak * max_peaks)()
    n = lib().orc_find_peaks(_p(cum, C.c_float), cum.size, cumulation_size, np.float32(threshold), sample_rate, center,
                             arr, max_peaks)
    return [arr[i].astuple() for i in range(min(n, max_peaks))]


def decode_iq_message(payload: bytes) -> np.ndarray:
    """kiwi/client.go:284-308: SND message body -> interleaved I,Q float32."""
    out = np.empty(max(0, (len(payload) - 17) // 2), np.float32)
    n = lib().orc_decode_iq_message(payload, len(payload), _p(out, C.c_float))
    if n < 0:
        raise ValueError("payload shorter than the 17-byte SND header")
    return out[:n]


class Decoder:
    """cw.Decoder (cw/decode.go:108-354) collecting its output as a str."""

    def __init__(self, sample_rate: int, block_size: int):
        self._h = lib().orc_decoder_new(sample_rate, block_size)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_decoder_free(self._h)
            self._h = None

    def reset(self):
        lib().orc_decoder_reset(self._h)

    def clear(self):
        lib().orc_decoder_clear(self._h)

    def preset_wpm(self, wpm: int):
        lib().orc_decoder_preset_wpm(self._h, wpm)

    def tick(self, state: bool):
        lib().orc_decoder_tick(self._h, int(bool(state)))

    def ticks(self, states):
        a = np.ascontiguousarray(states, dtype=np.uint8)
        lib().orc_decoder_ticks(self._h, _p(a, C.c_uint8), a.size)

    def stop(self):
        lib().orc_decoder_stop(self._h)

    @property
    def wpm(self) -> float:
        return lib().orc_decoder_wpm(self._h)

    def text(self) -> str:
        return runes_to_str(lib().orc_decoder_out(self._h), lib().orc_decoder_out_len(self._h))

    def buffer_reset(self):
        lib().orc_decoder_out_reset(self._h)

    def state(self) -> np.ndarray:
        out = np.empty(12)
        lib().orc_decoder_state(self._h, _p(out, C.c_double))
        return out


DEFAULT_TIMING = (1, 3, 1, 3, 7)  # cw/decode_test.go:233


def generate_stream(sample_rate: int, block_size: int, wpm: int, text: str, timing=DEFAULT_TIMING) -> np.ndarray:
    t = (C.c_int * 5)(*timing)
    runes = np.array([ord(ch) for ch in text], dtype=np.uint32)
    n = lib().orc_generate_stream(sample_rate, block_size, wpm, t, _p(runes, C.c_uint32), runes.size, None, 0)
    out = np.empty(n, np.uint8)
    lib().orc_generate_stream(sample_rate, block_size, wpm, t, _p(runes, C.c_uint32), runes.size, _p(out, C.c_uint8), n)
    return out


def morse_table() -> dict:
    L = lib()
    return {chr(L.orc_morse_rune(i)): L.orc_morse_code(i).decode() for i in range(L.orc_morse_count())}


class AudioDemodulator:
    """cw.AudioDemodulator (cw/audio.go) — mono float32 samples in, text out."""

    def __init__(self, pitch: float, sample_rate: int):
        self._h = lib().orc_audio_new(pitch, sample_rate)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_audio_free(self._h)
            self._h = None

    @property
    def blocksize(self) -> int:
        return lib().orc_audio_blocksize(self._h)

    @property
    def coeff(self) -> float:
        return lib().orc_audio_coeff(self._h)

    def set_scale(self, s: float):
        lib().orc_audio_set_scale(self._h, s)

    def set_debounce(self, t: int):
        lib().orc_audio_set_debounce(self._h, t)

    def set_magnitude_threshold(self, t: float):
        lib().orc_audio_set_magnitude_threshold(self._h, t)

    def write(self, samples: np.ndarray):
        s = np.ascontiguousarray(samples, dtype=np.float32)
        cap = s.size // self.blocksize + 2
        mags, raw, deb = np.empty(cap), np.empty(cap, np.uint8), np.empty(cap, np.uint8)
        n = lib().orc_audio_write(self._h, _p(s, C.c_float), s.size, _p(mags, C.c_double), _p(raw, C.c_uint8),
                                  _p(deb, C.c_uint8), cap)
        return mags[:n], raw[:n], deb[:n]

    def close(self):
        lib().orc_audio_close(self._h)

    def text(self) -> str:
        return runes_to_str(lib().orc_audio_out(self._h), lib().orc_audio_out_len(self._h))


class Receiver:
    """One band of rx.Receiver.run (rx/receiver.go:336-464) with an explicit listener set."""

    def __init__(self, sample_rate: int, block_size: int, edge_width: int, peak_threshold: float = 15.0,
                 debounce: int = 1, center_frequency: int = 0):
        self.n = block_size
        self.sample_rate = sample_rate
        self._h = lib().orc_receiver_new(sample_rate, block_size, edge_width, np.float32(peak_threshold), debounce,
                                         center_frequency)
        self.n_listeners = 0

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_receiver_free(self._h)
            self._h = None

    def attach(self, bin_: int) -> int:
        self.n_listeners += 1
        return lib().orc_receiver_attach(self._h, int(bin_))

    def detach(self, lid: int):
        lib().orc_receiver_detach(self._h, lid)

    def set_peak_threshold(self, t: float):
        lib().orc_receiver_set_peak_threshold(self._h, np.float32(t))

    def set_edge_width(self, e: int):
        lib().orc_receiver_set_edge_width(self._h, e)

    def set_find_peaks(self, on: bool):
        lib().orc_receiver_set_find_peaks(self._h, int(on))

    def text(self, lid: int) -> str:
        return runes_to_str(lib().orc_receiver_text(self._h, lid), lib().orc_receiver_text_len(self._h, lid))

    def decoder_state(self, lid: int) -> np.ndarray:
        out = np.empty(12)
        lib().orc_receiver_decoder_state(self._h, lid, _p(out, C.c_double))
        return out

    def cumulation(self):
        ptr = lib().orc_receiver_cumulation(self._h)
        return np.ctypeslib.as_array(ptr, shape=(self.n,)).copy(), lib().orc_receiver_cumulation_count(self._h)

    def process(self, iq: np.ndarray, want_spectrum: bool = False, max_peaks: int = 4096):
        """iq: float32 [n_frames, 2N].  Returns a dict of per-frame / per-listener / per-chunk outputs."""
        iq = np.ascontiguousarray(iq, dtype=np.float32).reshape(-1, 2 * self.n)
        F, Ls = iq.shape[0], self.n_listeners
        recs = np.zeros(F, FRAME_REC_DTYPE)
        spec = np.empty((F, self.n), np.float32) if want_spectrum else None
        psd = np.empty((F, self.n), np.float32) if want_spectrum else None
        vals = np.zeros((F, max(Ls, 1)), np.float32)
        raw = np.zeros((F, max(Ls, 1)), np.uint8)
        deb = np.zeros((F, max(Ls, 1)), np.uint8)
        max_chunks = F // 100 + 2
        peaks = (Peak * (max_chunks * max_peaks))()
        counts = np.zeros(max_chunks, np.int32)
        frames = np.zeros(max_chunks, np.int32)
        cum = np.zeros((max_chunks, self.n), np.float32)
        nch = lib().orc_receiver_process(
            self._h, _p(iq, C.c_float), F, recs.ctypes.data_as(C.POINTER(FrameRec)), _p(spec, C.c_float),
            _p(psd, C.c_float), _p(vals, C.c_float) if Ls else None, _p(raw, C.c_uint8) if Ls else None,
            _p(deb, C.c_uint8) if Ls else None, peaks, _p(counts, C.c_int), _p(frames, C.c_int), max_chunks, max_peaks,
            _p(cum, C.c_float))
        chunk_peaks = []
        for c in range(nch):
            chunk_peaks.append([peaks[c * max_peaks + i].astuple() for i in range(min(counts[c], max_peaks))])
        return {
            "frames": recs, "spectrum": spec, "psd": psd, "values": vals[:, :Ls], "raw": raw[:, :Ls],
            "deb": deb[:, :Ls], "n_chunks": nch, "peaks": chunk_peaks, "peak_frames": frames[:nch].copy(),
            "cumulation": cum[:nch],
        }

    def run_baseline(self, iq: np.ndarray) -> int:
        iq = np.ascontiguousarray(iq, dtype=np.float32)
        return lib().orc_receiver_run_baseline(self._h, _p(iq, C.c_float), iq.size // (2 * self.n))


class PeaksTable:
    """rx.PeaksTable (rx/peaks.go) with a manual clock (rx/receiver.go:41-55)."""

    NONE, NEW, ACTIVE, INACTIVE = 0, 1, 2, 3

    def __init__(self, size: int):
        self._h = lib().orc_peaks_table_new(size)
        self.now = 0.0

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_peaks_table_free(self._h)
            self._h = None

    def set_now(self, t):
        self.now = t
        lib().orc_peaks_table_set_now(self._h, t)

    def put(self, f, t):
        return lib().orc_peaks_table_put(self._h, f, t)

    def force_put(self, f, t):
        return lib().orc_peaks_table_force_put(self._h, f, t)

    def place(self, f, t, state):
        return lib().orc_peaks_table_place(self._h, f, t, state)

    def at(self, b):
        return lib().orc_peaks_table_at(self._h, b)

    def state(self, e):
        return lib().orc_peaks_table_state(self._h, e)

    def cleanup(self):
        lib().orc_peaks_table_cleanup(self._h)

    def activate(self, f, t):
        lib().orc_peaks_table_activate(self._h, f, t)

    def deactivate(self, f, t):
        lib().orc_peaks_table_deactivate(self._h, f, t)

    def find_next(self):
        return lib().orc_peaks_table_find_next(self._h)

    def entry(self, e):
        a, b = C.c_int(), C.c_int()
        lib().orc_peaks_table_entry(self._h, e, C.byref(a), C.byref(b))
        return a.value, b.value
