"""ctypes binding of libsdrainer_hip.so — the same C ABI (include/sdrainer_hip.h) a cgo shim binds.

There is no CPU implementation behind this module: if the HIP library is missing or fails to load,
importing raises.  PyTorch is only used by callers for device memory / streams; the ABI takes raw
device pointers.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SDR_HIP_LIB") or os.path.join(_HERE, "csrc", "libsdrainer_hip.so")  # env: diagnostic builds

OK, ERR_BAD_ARG, ERR_BAD_RATE, ERR_BAD_SIZE, ERR_WOULD_DROP, ERR_HIP, ERR_NO_SLOT, ERR_STATE, ERR_WOULD_BLOCK = range(9)
CUMULATION_SIZE = 100
KERNELS = ("k_fft_psd", "k_window_means", "k_noise_stats", "k_thresholds", "k_listen_gather", "k_cumulate",
           "k_find_peaks", "k_listen_decode")


class SdrError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libsdrainer_hip status {code}: {msg}")
        self.code = code


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("n_bands", C.c_int32), ("sample_rate", C.c_int32), ("block_size", C.c_int32),
        ("edge_width", C.c_int32), ("peak_threshold", C.c_float), ("signal_debounce", C.c_int32),
        ("max_listeners", C.c_int32), ("max_batch_frames", C.c_int32), ("max_peaks", C.c_int32),
        ("find_peaks", C.c_int32), ("trace", C.c_int32), ("device_id", C.c_int32), ("reserved", C.c_int32),
    ]


class Peak(C.Structure):
    _fields_ = [
        ("from_", C.c_int32), ("to", C.c_int32), ("from_frequency", C.c_int64), ("to_frequency", C.c_int64),
        ("signal_frequency", C.c_int64), ("signal_value", C.c_float), ("signal_bin", C.c_int32),
    ]

    def astuple(self):
        return (self.from_, self.to, self.from_frequency, self.to_frequency, self.signal_frequency,
                float(np.float32(self.signal_value)), self.signal_bin)


FRAME_REC_DTYPE = np.dtype([("min_mean", "<f4"), ("dev_in", "<f4"), ("variance", "<f8"), ("nf_in", "<f4"),
                            ("noise_dev", "<f4"), ("noise_floor", "<f4"), ("peak_thr", "<f4"), ("listen_thr", "<f4"),
                            ("pad", "<f4")])
EDGE_DTYPE = np.dtype([("frame", "<u4"), ("state", "<u4")])
PEAK_DTYPE = np.dtype([("from", "<i4"), ("to", "<i4"), ("from_frequency", "<i8"), ("to_frequency", "<i8"),
                       ("signal_frequency", "<i8"), ("signal_value", "<f4"), ("signal_bin", "<i4")])
CHUNK_RESULT_DTYPE = np.dtype([("band", "<i4"), ("n_peaks", "<i4"), ("frame", "<i8"), ("first_peak", "<i4"),
                               ("peaks_found", "<i4")])
LISTENER_RESULT_DTYPE = np.dtype([("band", "<i4"), ("listener", "<i4"), ("first_edge", "<i4"), ("n_edges", "<i4"),
                                  ("first_rune", "<i4"), ("n_runes", "<i4")])


class ScopeSpectralFrame(C.Structure):
    """sdr_scope_spectral_frame: scope.SpectralFrame without its Values (scope/scope.go:24-31)."""
    _fields_ = [("frame", C.c_int64), ("from_frequency", C.c_double), ("to_frequency", C.c_double),
                ("signal_bin", C.c_double), ("threshold", C.c_double), ("n_values", C.c_int32), ("reserved", C.c_int32)]


SCOPE_TIME_DTYPE = np.dtype([("threshold", "<f8"), ("value", "<f8"), ("state", "<f8"), ("debounced", "<f8")])
SCOPE_DECODE_DTYPE = np.dtype([("frame", "<i8"), ("duration", "<f8"), ("state", "<f8"), ("on_threshold", "<f8"),
                               ("on_threshold_low", "<f8"), ("on_threshold_high", "<f8"), ("off_threshold", "<f8"),
                               ("off_threshold_low", "<f8"), ("off_threshold_high", "<f8")])


class Results(C.Structure):
    """sdr_results (include/sdrainer_hip.h)."""
    _fields_ = [
        ("struct_size", C.c_int32), ("n_frames", C.c_int32), ("batch_index", C.c_int64), ("first_frame", C.c_int64),
        ("chunks", C.c_void_p), ("chunks_cap", C.c_int32), ("n_chunks", C.c_int32),
        ("peaks", C.c_void_p), ("peaks_cap", C.c_int32), ("n_peaks", C.c_int32),
        ("listeners", C.c_void_p), ("listeners_cap", C.c_int32), ("n_listeners", C.c_int32),
        ("edges", C.c_void_p), ("edges_cap", C.c_int32), ("n_edges", C.c_int32),
        ("runes", C.c_void_p), ("rune_frames", C.c_void_p), ("runes_cap", C.c_int32), ("n_runes", C.c_int32),
        ("runes_dropped", C.c_uint64), ("edges_dropped", C.c_uint64),
    ]

_lib = None

# every symbol include/sdrainer_hip.h declares (tests/test_capi_symbols.py checks the header against this)
SYMBOLS = (
    "sdr_last_error sdr_abi_version sdr_create sdr_destroy sdr_self_check sdr_set_stream sdr_push_iq sdr_push_kiwi_snd sdr_staged_frames "
    "sdr_process_staged sdr_process_staged_limit sdr_process_device sdr_sync sdr_attach sdr_detach sdr_listener_count sdr_listener_stop "
    "sdr_set_peak_threshold sdr_set_edge_width sdr_set_signal_debounce sdr_set_center_frequency sdr_set_find_peaks "
    "sdr_last_batch_frames sdr_total_frames sdr_last_batch_chunks sdr_read_peaks sdr_read_cumulation sdr_read_text "
    "sdr_read_edges sdr_read_keying_bits sdr_read_frame_records sdr_read_trace sdr_read_spectrum "
    "sdr_read_decoder_state sdr_graph_batches sdr_graph_capture sdr_graph_launch sdr_graph_release sdr_scope_active sdr_scope_read_spectral sdr_scope_read_demod sdr_scope_read_decode sdr_enable_results sdr_poll sdr_defer_listen sdr_listen_pending sdr_poll_peaks sdr_attach_at sdr_process_listen sdr_results_pending sdr_read_drop_counters sdr_profile_enable sdr_profile_read sdr_profile_reset sdr_kernel_name sdr_audio_create "
    "sdr_audio_destroy sdr_audio_blocksize sdr_audio_set_scale sdr_audio_set_debounce "
    "sdr_audio_set_magnitude_threshold sdr_audio_write sdr_audio_close sdr_audio_read_text sdr_audio_read_trace"
).split()


def load():
    """Loads the HIP library.  Fails loudly: there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m sdrainer_amd.csrc.build` (hipcc, gfx950). "
            "sdrainer_amd has no CPU fallback.")
    # PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64.so.1.  Two HSA runtimes
    # in one process fight over the device ("No HIP GPUs are available" for whichever comes second),
    # so when torch is installed let it load its copy first: the dynamic loader then resolves this
    # library's libamdhip64.so.7 to the already-loaded one.  Without torch (e.g. under a cgo host) the
    # system ROCm runtime is used.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, ip, fp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float)

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)

    sig("sdr_last_error", C.c_char_p)
    sig("sdr_abi_version", C.c_int)
    sig("sdr_create", C.c_int, C.POINTER(Config), C.POINTER(vp))
    sig("sdr_self_check", C.c_int, C.c_int)
    sig("sdr_destroy", C.c_int, vp)
    sig("sdr_set_stream", C.c_int, vp, vp)
    sig("sdr_push_iq", C.c_int, vp, C.c_int, C.c_int, fp, C.c_size_t)
    sig("sdr_push_kiwi_snd", C.c_int, vp, C.c_int, C.c_int, C.c_char_p, C.c_size_t)
    sig("sdr_staged_frames", C.c_int, vp, C.c_int)
    sig("sdr_process_staged", C.c_int, vp, ip)
    sig("sdr_process_staged_limit", C.c_int, vp, C.c_int, ip)
    sig("sdr_process_device", C.c_int, vp, vp, C.c_int)
    sig("sdr_sync", C.c_int, vp)
    sig("sdr_attach", C.c_int, vp, C.c_int, C.c_int, ip)
    sig("sdr_detach", C.c_int, vp, C.c_int, C.c_int)
    sig("sdr_listener_count", C.c_int, vp, C.c_int)
    sig("sdr_listener_stop", C.c_int, vp, C.c_int, C.c_int)
    sig("sdr_set_peak_threshold", C.c_int, vp, C.c_int, C.c_float)
    sig("sdr_set_edge_width", C.c_int, vp, C.c_int)
    sig("sdr_set_signal_debounce", C.c_int, vp, C.c_int, C.c_int)
    sig("sdr_set_center_frequency", C.c_int, vp, C.c_int, C.c_int64)
    sig("sdr_set_find_peaks", C.c_int, vp, C.c_int)
    sig("sdr_last_batch_frames", C.c_int, vp)
    sig("sdr_total_frames", C.c_int64, vp)
    sig("sdr_last_batch_chunks", C.c_int, vp)
    sig("sdr_read_peaks", C.c_int, vp, C.c_int, C.c_int, C.POINTER(Peak), C.c_int, ip, ip)
    sig("sdr_read_cumulation", C.c_int, vp, C.c_int, C.c_int, fp)
    sig("sdr_read_text", C.c_int, vp, C.c_int, C.c_int, C.c_char_p, C.c_int, ip)
    sig("sdr_read_edges", C.c_int, vp, C.c_int, C.c_int, vp, C.c_int, ip)
    sig("sdr_read_keying_bits", C.c_int, vp, C.c_int, C.c_int, vp, C.c_int)
    sig("sdr_read_frame_records", C.c_int, vp, C.c_int, vp, C.c_int)
    sig("sdr_read_trace", C.c_int, vp, C.c_int, C.c_int, vp, vp, vp, C.c_int)
    sig("sdr_read_spectrum", C.c_int, vp, C.c_int, C.c_int, vp, vp)
    sig("sdr_read_decoder_state", C.c_int, vp, C.c_int, C.c_int, C.POINTER(C.c_double))
    sig("sdr_graph_batches", C.c_int, vp)
    sig("sdr_graph_capture", C.c_int, vp, C.c_int)
    sig("sdr_graph_launch", C.c_int, vp, C.POINTER(C.c_void_p))
    sig("sdr_graph_release", C.c_int, vp)
    sig("sdr_scope_active", C.c_int, vp)
    sig("sdr_scope_read_spectral", C.c_int, vp, C.c_int, C.c_int, C.POINTER(ScopeSpectralFrame), C.POINTER(C.c_double), C.c_int)
    sig("sdr_scope_read_demod", C.c_int, vp, C.c_int, C.c_int, vp, C.c_int, ip)
    sig("sdr_scope_read_decode", C.c_int, vp, C.c_int, C.c_int, vp, C.c_int, ip)
    sig("sdr_enable_results", C.c_int, vp, C.c_int)
    sig("sdr_poll", C.c_int, vp, C.POINTER(Results), C.c_int)
    sig("sdr_defer_listen", C.c_int, vp, C.c_int)
    sig("sdr_listen_pending", C.c_int, vp)
    sig("sdr_poll_peaks", C.c_int, vp, C.POINTER(Results), C.c_int)
    sig("sdr_attach_at", C.c_int, vp, C.c_int, C.c_int, C.c_int64, ip)
    sig("sdr_process_listen", C.c_int, vp)
    sig("sdr_results_pending", C.c_int, vp)
    sig("sdr_read_drop_counters", C.c_int, vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64))
    sig("sdr_profile_enable", C.c_int, vp, C.c_int)
    sig("sdr_profile_read", C.c_int, vp, C.c_int, C.POINTER(C.c_double), ip)
    sig("sdr_profile_reset", C.c_int, vp)
    sig("sdr_kernel_name", C.c_char_p, C.c_int)
    sig("sdr_audio_create", C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.POINTER(vp))
    sig("sdr_audio_destroy", C.c_int, vp)
    sig("sdr_audio_blocksize", C.c_int, vp)
    sig("sdr_audio_set_scale", C.c_int, vp, C.c_double)
    sig("sdr_audio_set_debounce", C.c_int, vp, C.c_int)
    sig("sdr_audio_set_magnitude_threshold", C.c_int, vp, C.c_double)
    sig("sdr_audio_write", C.c_int, vp, fp, C.c_int)
    sig("sdr_audio_close", C.c_int, vp)
    sig("sdr_audio_read_text", C.c_int, vp, C.c_int, C.c_char_p, C.c_int, ip)
    sig("sdr_audio_read_trace", C.c_int, vp, C.c_int, vp, vp, vp, C.c_int, ip)
    _lib = L
    return L


def _check(rc: int):
    if rc != OK:
        raise SdrError(rc, load().sdr_last_error().decode(errors="replace"))


def _vp(a):
    return C.c_void_p(a.ctypes.data) if a is not None else None


class Bank:
    """A bank of n_bands receivers on one GPU (see include/sdrainer_hip.h)."""

    def __init__(self, sample_rate: int, block_size: int, n_bands: int = 1, edge_width: int | None = None,
                 peak_threshold: float = 15.0, signal_debounce: int = 1, max_listeners: int = 30,
                 max_batch_frames: int = 1024, max_peaks: int = 1024, find_peaks: bool = True, trace: bool = False,
                 device_id: int = 0):
        L = load()
        if edge_width is None:
            edge_width = 70 * block_size // 512  # the reference default (rx/receiver.go:25) scaled with N
        self.cfg = Config(C.sizeof(Config), n_bands, sample_rate, block_size, edge_width, peak_threshold,
                          signal_debounce, max_listeners, max_batch_frames, max_peaks, int(find_peaks), int(trace),
                          device_id, 0)
        self.n = block_size
        self.n_bands = n_bands
        h = C.c_void_p()
        _check(L.sdr_create(C.byref(self.cfg), C.byref(h)))
        self._h = h
        self._L = L

    def close(self):
        if getattr(self, "_h", None):
            self._L.sdr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # producer ---------------------------------------------------------------------------------
    def set_stream(self, stream_ptr: int):
        _check(self._L.sdr_set_stream(self._h, C.c_void_p(stream_ptr)))

    def push_iq(self, band: int, sample_rate: int, iq: np.ndarray) -> int:
        """Returns the status code (0 ok) instead of raising for the reference's log-and-drop cases."""
        iq = np.ascontiguousarray(iq, dtype=np.float32)
        rc = self._L.sdr_push_iq(self._h, band, sample_rate, iq.ctypes.data_as(C.POINTER(C.c_float)), iq.size)
        if rc not in (OK, ERR_BAD_RATE, ERR_BAD_SIZE, ERR_WOULD_DROP, ERR_STATE):
            _check(rc)
        return rc

    def push_kiwi_snd(self, band: int, sample_rate: int, payload: bytes) -> int:
        """payload: body of one KiwiSDR SND message (17-byte header + big-endian int16 IQ)."""
        rc = self._L.sdr_push_kiwi_snd(self._h, band, sample_rate, payload, len(payload))
        if rc not in (OK, ERR_BAD_RATE, ERR_BAD_SIZE, ERR_WOULD_DROP, ERR_STATE):
            _check(rc)
        return rc

    def staged_frames(self, band: int) -> int:
        return self._L.sdr_staged_frames(self._h, band)

    def process_staged(self) -> int:
        n = C.c_int()
        _check(self._L.sdr_process_staged(self._h, C.byref(n)))
        return n.value

    def process_staged_limit(self, max_frames: int) -> int:
        n = C.c_int()
        _check(self._L.sdr_process_staged_limit(self._h, max_frames, C.byref(n)))
        return n.value

    def process_device(self, iq_dev_ptr: int, n_frames: int):
        _check(self._L.sdr_process_device(self._h, C.c_void_p(iq_dev_ptr), n_frames))

    def process_host(self, iq: np.ndarray) -> int:
        """iq: float32 [n_bands, n_frames, 2N] (or [n_frames, 2N] for one band) from host memory."""
        iq = np.ascontiguousarray(iq, dtype=np.float32).reshape(self.n_bands, -1, 2 * self.n)
        for b in range(self.n_bands):
            rc = self.push_iq(b, self.cfg.sample_rate, iq[b])
            if rc != OK:
                raise SdrError(rc, self._L.sdr_last_error().decode())
        return self.process_staged()

    def sync(self):
        _check(self._L.sdr_sync(self._h))

    # listeners --------------------------------------------------------------------------------
    def attach(self, band: int, bin_: int) -> int:
        lid = C.c_int()
        _check(self._L.sdr_attach(self._h, band, int(bin_), C.byref(lid)))
        return lid.value

    def attach_at(self, band: int, bin_: int, start_frame: int) -> int:
        """sdr_attach_at: a listener that listens from bank frame `start_frame` of the batch waiting for its listen half."""
        lid = C.c_int(-1)
        _check(self._L.sdr_attach_at(self._h, band, int(bin_), int(start_frame), C.byref(lid)))
        return lid.value

    def defer_listen(self, on: bool = True) -> None:
        _check(self._L.sdr_defer_listen(self._h, int(on)))

    @property
    def listen_pending(self) -> bool:
        return bool(self._L.sdr_listen_pending(self._h))

    def process_listen(self) -> None:
        _check(self._L.sdr_process_listen(self._h))

    def detach(self, band: int, lid: int):
        _check(self._L.sdr_detach(self._h, band, lid))

    def listener_count(self, band: int) -> int:
        return self._L.sdr_listener_count(self._h, band)

    def listener_stop(self, band: int, lid: int):
        _check(self._L.sdr_listener_stop(self._h, band, lid))

    # control ----------------------------------------------------------------------------------
    def set_peak_threshold(self, band: int, t: float):
        _check(self._L.sdr_set_peak_threshold(self._h, band, t))

    def set_edge_width(self, e: int):
        _check(self._L.sdr_set_edge_width(self._h, e))

    def set_signal_debounce(self, band: int, d: int):
        _check(self._L.sdr_set_signal_debounce(self._h, band, d))

    def set_center_frequency(self, band: int, f: int):
        _check(self._L.sdr_set_center_frequency(self._h, band, f))

    def set_find_peaks(self, on: bool):
        _check(self._L.sdr_set_find_peaks(self._h, int(on)))

    # consumer ---------------------------------------------------------------------------------
    @property
    def last_batch_frames(self) -> int:
        return self._L.sdr_last_batch_frames(self._h)

    @property
    def total_frames(self) -> int:
        return self._L.sdr_total_frames(self._h)

    @property
    def last_batch_chunks(self) -> int:
        return self._L.sdr_last_batch_chunks(self._h)

    def read_peaks(self, band: int, chunk: int):
        cap = self.cfg.max_peaks
        arr = (Peak * cap)()
        n, fr = C.c_int(), C.c_int()
        _check(self._L.sdr_read_peaks(self._h, band, chunk, arr, cap, C.byref(n), C.byref(fr)))
        return [arr[i].astuple() for i in range(min(n.value, cap))], n.value, fr.value

    def read_cumulation(self, band: int, chunk: int) -> np.ndarray:
        out = np.empty(self.n, np.float32)
        _check(self._L.sdr_read_cumulation(self._h, band, chunk, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def read_text(self, band: int, lid: int) -> str:
        buf = C.create_string_buffer(16384)
        n = C.c_int()
        _check(self._L.sdr_read_text(self._h, band, lid, buf, len(buf), C.byref(n)))
        return buf.raw[:n.value].decode("utf-8")

    def read_edges(self, band: int, lid: int) -> np.ndarray:
        cap = min(self.cfg.max_batch_frames, 8192)
        out = np.zeros(cap, EDGE_DTYPE)
        n = C.c_int()
        _check(self._L.sdr_read_edges(self._h, band, lid, _vp(out), cap, C.byref(n)))
        return out[:min(n.value, cap)]

    def read_keying_bits(self, band: int, lid: int) -> np.ndarray:
        """Debounced on/off state per frame of the last batch, as uint8 [n_frames]."""
        nf = self.last_batch_frames
        words = (nf + 63) // 64
        out = np.zeros(max(words, 1), np.uint64)
        _check(self._L.sdr_read_keying_bits(self._h, band, lid, _vp(out), words))
        bits = np.unpackbits(out.view(np.uint8), bitorder="little")
        return bits[:nf].astype(np.uint8)

    def read_frame_records(self, band: int) -> np.ndarray:
        nf = self.last_batch_frames
        out = np.zeros(max(nf, 1), FRAME_REC_DTYPE)
        _check(self._L.sdr_read_frame_records(self._h, band, _vp(out), nf))
        return out[:nf]

    def read_trace(self, band: int, lid: int):
        nf = self.last_batch_frames
        v, r, d = np.zeros(nf, np.float32), np.zeros(nf, np.uint8), np.zeros(nf, np.uint8)
        _check(self._L.sdr_read_trace(self._h, band, lid, _vp(v), _vp(r), _vp(d), nf))
        return v, r, d

    def read_spectrum(self, band: int, frame: int):
        sp, psd = np.empty(self.n, np.float32), np.empty(self.n, np.float32)
        _check(self._L.sdr_read_spectrum(self._h, band, frame, _vp(sp), _vp(psd)))
        return sp, psd

    def read_decoder_state(self, band: int, lid: int) -> np.ndarray:
        out = np.empty(12)
        _check(self._L.sdr_read_decoder_state(self._h, band, lid, out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    # graph mode -------------------------------------------------------------------------------
    @property
    def graph_batches(self) -> int:
        return self._L.sdr_graph_batches(self._h)

    def graph_capture(self, n_frames: int):
        _check(self._L.sdr_graph_capture(self._h, n_frames))

    def graph_launch(self, iq_dev_ptrs):
        arr = (C.c_void_p * len(iq_dev_ptrs))(*[C.c_void_p(int(p)) for p in iq_dev_ptrs])
        assert len(iq_dev_ptrs) == self.graph_batches
        _check(self._L.sdr_graph_launch(self._h, arr))

    def graph_release(self):
        _check(self._L.sdr_graph_release(self._h))

    # scope tap --------------------------------------------------------------------------------
    @property
    def scope_active(self) -> bool:
        return bool(self._L.sdr_scope_active(self._h))

    def scope_spectral_frame(self, band: int, chunk: int):
        """(header dict, Values float64[N]) of the "spectrum" scope stream for one completed cumulation."""
        fr = ScopeSpectralFrame()
        vals = np.empty(self.n, np.float64)
        _check(self._L.sdr_scope_read_spectral(self._h, band, chunk, C.byref(fr), vals.ctypes.data_as(C.POINTER(C.c_double)),
                                               self.n))
        hdr = {"frame": fr.frame, "from_frequency": fr.from_frequency, "to_frequency": fr.to_frequency,
               "signal_bin": fr.signal_bin, "threshold": fr.threshold, "n_values": fr.n_values}
        return hdr, vals

    def scope_demod_frames(self, band: int, lid: int) -> np.ndarray:
        """The listener's "demod" scope stream of the last batch: threshold, value, state, debounced per frame."""
        nf = self.last_batch_frames
        out = np.zeros(max(nf, 1), SCOPE_TIME_DTYPE)
        n = C.c_int()
        _check(self._L.sdr_scope_read_demod(self._h, band, lid, _vp(out), nf, C.byref(n)))
        return out[:min(nf, n.value)]

    def scope_decode_frames(self, band: int, lid: int) -> np.ndarray:
        """cw.Decoder's scope streams of the last batch (sdr_scope_decode_frame), one record per tick the listener took."""
        nf = self.last_batch_frames
        out = np.zeros(max(nf, 1), SCOPE_DECODE_DTYPE)
        n = C.c_int()
        _check(self._L.sdr_scope_read_decode(self._h, band, lid, _vp(out), nf, C.byref(n)))
        return out[:min(nf, n.value)]

    # bulk delivery ----------------------------------------------------------------------------
    def enable_results(self, on: bool = True):
        _check(self._L.sdr_enable_results(self._h, int(on)))
        if on and not hasattr(self, "_res"):
            c = self.cfg
            chunks = (c.max_batch_frames // CUMULATION_SIZE + 2) * c.n_bands
            listeners = max(c.max_listeners * c.n_bands, 1)
            self._res_bufs = {
                "chunks": np.zeros(chunks, CHUNK_RESULT_DTYPE),
                "peaks": np.zeros(chunks * c.max_peaks, PEAK_DTYPE),
                "listeners": np.zeros(listeners, LISTENER_RESULT_DTYPE),
                "edges": np.zeros(listeners * min(c.max_batch_frames, 8192), EDGE_DTYPE),
                "runes": np.zeros(listeners * 2048, np.uint32),
            }
            self._rune_frames = np.zeros(listeners * 2048, np.uint32)
            r = Results()
            r.struct_size = C.sizeof(Results)
            for k, a in self._res_bufs.items():
                setattr(r, k, a.ctypes.data)
                setattr(r, k + "_cap", len(a))
            r.rune_frames = self._rune_frames.ctypes.data
            self._res = r

    def poll_peaks(self, wait: bool = True, copy: bool = True):
        """sdr_poll_peaks: chunks and peaks of the batch that waits for its listen half (it stays undelivered)."""
        return self.poll(wait, copy, _entry=self._L.sdr_poll_peaks)

    def poll(self, wait: bool = False, copy: bool = True, _entry=None):
        """Oldest finished batch as a dict of record arrays (views into reused buffers unless `copy`), or None."""
        r = self._res
        rc = (_entry or self._L.sdr_poll)(self._h, C.byref(r), int(wait))
        if rc == ERR_WOULD_BLOCK:
            return None
        _check(rc)
        out = {"batch_index": r.batch_index, "first_frame": r.first_frame, "n_frames": r.n_frames,
               "runes_dropped": r.runes_dropped, "edges_dropped": r.edges_dropped}
        for k, a in self._res_bufs.items():
            v = a[:getattr(r, "n_" + k)]
            out[k] = v.copy() if copy else v
        v = self._rune_frames[:r.n_runes]
        out["rune_frames"] = v.copy() if copy else v
        return out

    def poll_counts(self, wait: bool = False):
        """Like poll() but returns only the record counts (the bench's timed loop: no Python-side copies)."""
        r = self._res
        rc = self._L.sdr_poll(self._h, C.byref(r), int(wait))
        if rc == ERR_WOULD_BLOCK:
            return None
        _check(rc)
        return (r.batch_index, r.n_chunks, r.n_peaks, r.n_listeners, r.n_edges, r.n_runes, r.runes_dropped, r.edges_dropped)

    @property
    def results_pending(self) -> int:
        return self._L.sdr_results_pending(self._h)

    def read_drop_counters(self):
        a, b = C.c_uint64(), C.c_uint64()
        _check(self._L.sdr_read_drop_counters(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    @staticmethod
    def runes_to_text(runes) -> str:
        return "".join(chr(int(x)) for x in runes)

    # measurement ------------------------------------------------------------------------------
    def profile_enable(self, on: bool):
        _check(self._L.sdr_profile_enable(self._h, int(on)))

    def profile_reset(self):
        _check(self._L.sdr_profile_reset(self._h))

    def profile_read(self) -> dict:
        out = {}
        for k, name in enumerate(KERNELS):
            ms, n = C.c_double(), C.c_int()
            _check(self._L.sdr_profile_read(self._h, k, C.byref(ms), C.byref(n)))
            out[name] = (ms.value, n.value)
        return out


class AudioBank:
    """n_streams cw.AudioDemodulator instances (cw/audio.go) on the GPU."""

    def __init__(self, n_streams: int, pitch: float, sample_rate: int, max_blocks: int = 4096, device_id: int = 0):
        L = load()
        h = C.c_void_p()
        _check(L.sdr_audio_create(n_streams, pitch, sample_rate, max_blocks, device_id, C.byref(h)))
        self._h, self._L, self.n_streams = h, L, n_streams

    def close_handle(self):
        if getattr(self, "_h", None):
            self._L.sdr_audio_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close_handle()
        except Exception:
            pass

    @property
    def blocksize(self) -> int:
        return self._L.sdr_audio_blocksize(self._h)

    def set_scale(self, s: float):
        _check(self._L.sdr_audio_set_scale(self._h, s))

    def set_debounce(self, t: int):
        _check(self._L.sdr_audio_set_debounce(self._h, t))

    def set_magnitude_threshold(self, t: float):
        _check(self._L.sdr_audio_set_magnitude_threshold(self._h, t))

    def write(self, samples: np.ndarray):
        s = np.ascontiguousarray(samples, dtype=np.float32).reshape(self.n_streams, -1)
        _check(self._L.sdr_audio_write(self._h, s.ctypes.data_as(C.POINTER(C.c_float)), s.shape[1]))

    def close(self):
        _check(self._L.sdr_audio_close(self._h))

    def read_text(self, stream: int) -> str:
        buf = C.create_string_buffer(65536)
        n = C.c_int()
        _check(self._L.sdr_audio_read_text(self._h, stream, buf, len(buf), C.byref(n)))
        return buf.raw[:n.value].decode("utf-8")

    def read_trace(self, stream: int, max_blocks: int = 1 << 16):
        m, r, d = np.zeros(max_blocks), np.zeros(max_blocks, np.uint8), np.zeros(max_blocks, np.uint8)
        n = C.c_int()
        _check(self._L.sdr_audio_read_trace(self._h, stream, _vp(m), _vp(r), _vp(d), max_blocks, C.byref(n)))
        k = min(n.value, max_blocks)
        return m[:k], r[:k], d[:k]
