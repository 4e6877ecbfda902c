/*
 * sdrainer_hip.h — C ABI of libsdrainer_hip.so, the MI355X (gfx950) implementation of sdrainer's
 * per-block IQ-strainer DSP.  This is the drop-in boundary: plain pointers and sizes, no C++ or
 * framework types.  A Go maintainer binds it with cgo behind rx.Receiver / rx.Listener
 * (INTEGRATION.md shows the stub); this repo's C++ host mirror (sdrainer_amd/csrc/host/) and its
 * Python tests bind exactly the same symbols.
 *
 * One `sdr_bank` = n_bands independent receivers of identical geometry (sample rate, block size) on
 * one GPU and one HIP stream.  A band is what the reference calls a Receiver: it owns its rolling
 * noise floor, cumulation, listeners and decoders and shares nothing with other bands
 * (rx/receiver.go:64-91), which is what makes bands shardable across GPUs.
 *
 * Reference interface each entry point replaces (paths relative to the reference checkout):
 *   sdr_create / sdr_destroy        rx.NewReceiver + Receiver.Start / Stop      rx/receiver.go:93,130,148
 *   sdr_push_iq                     Receiver.IQData(sampleRate, []float32)      rx/receiver.go:315-334
 *   sdr_process_staged              the frame case of Receiver.run              rx/receiver.go:353-463
 *   sdr_process_device              same, for IQ already resident in HBM (the "IQ ring buffer")
 *   sdr_attach / sdr_detach         ListenerPool.BindNext + Listener.Attach / Detach + Release
 *                                                                               rx/listener.go:84-108,214-248
 *   sdr_set_peak_threshold          Receiver.SetPeakThreshold                   rx/receiver.go:208-212
 *   sdr_set_edge_width              Receiver.SetEdgeWidth                       rx/receiver.go:214-218
 *   sdr_set_signal_debounce         Receiver.SetSignalDebounce                  rx/receiver.go:238-244
 *   sdr_set_center_frequency        Receiver.SetCenterFrequency                 rx/receiver.go:246-253
 *   sdr_read_peaks                  the []dsp.Peak of dsp.FindPeaks             dsp/fft.go:179-188,254-285
 *   sdr_read_text                   the io.Writer each Listener's Decoder writes to  cw/decode.go:352-355
 *   sdr_read_edges / _trace         what cw.SpectralDemodulator.Tick hands to Decoder.Tick  cw/spectral.go:48-54
 *   sdr_read_frame_records          locals of Receiver.run (noise floor, thresholds)  rx/receiver.go:381-385
 *   sdr_push_kiwi_snd               decodeIQMessage + kiwi.Process.IQData       kiwi/client.go:284-308, kiwi/kiwi.go:94-105
 *   sdr_audio_*                     cw.AudioDemodulator (Goertzel audio path)   cw/audio.go:37-211
 *   sdr_enable_results / sdr_poll   the consumer side in bulk: what Receiver.run hands to its listeners'
 *                                   io.Writer (rx/receiver.go:123, ChannelWriter :508-539) and to the
 *                                   Reporter (rx/rx.go:11-17), one call per processed batch
 *   sdr_defer_listen / sdr_poll_peaks / sdr_attach_at / sdr_process_listen
 *                                   the discovery branch of Receiver.run - FindPeaks at a cumulation boundary,
 *                                   PeaksTable.FindNext, ListenerPool.BindNext, Listener.Attach, the listener
 *                                   hearing the next frame - without one host round trip per cumulation
 *                                                                               rx/receiver.go:404-426
 *   sdr_scope_*                     scope.Scope.ShowSpectralFrame / ShowTimeFrame  scope/scope.go:14-37,
 *                                   call sites rx/receiver.go:428-457, cw/spectral.go:56-81, cw/decode.go:228-243
 *
 * Semantics kept from the reference: setters take effect between frames, never mid-frame (here: at
 * the next process call, rx/receiver.go:166-172); wrong sample rate / block size / a full queue do
 * not abort anything, the call returns a status the shim maps to the reference's log-and-drop
 * (rx/receiver.go:319-333).  The one semantic extension is batching: a process call consumes many
 * frames per band, in order.
 *
 * Threading: a bank is single-producer (like the reference's run goroutine, all DSP state is owned
 * by one thread); different banks are independent.  One more thread may consume: sdr_poll and
 * sdr_results_pending may be called from a thread of their own while the producer thread processes (the
 * reference's Reporter callbacks arrive on other goroutines too).
 */
#ifndef SDRAINER_HIP_H
#define SDRAINER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDR_ABI_VERSION 2

/* status codes */
#define SDR_OK 0
#define SDR_ERR_BAD_ARG 1    /* null pointer, index out of range, unsupported geometry            */
#define SDR_ERR_BAD_RATE 2   /* wrong incoming sample rate  -> reference logs + drops (:319-322)  */
#define SDR_ERR_BAD_SIZE 3   /* wrong incoming block size   -> reference logs + drops (:323-326)  */
#define SDR_ERR_WOULD_DROP 4 /* staging queue full          -> reference logs + drops (:328-333)  */
#define SDR_ERR_HIP 5        /* a HIP runtime call failed; sdr_last_error() has the text          */
#define SDR_ERR_NO_SLOT 6    /* listener pool exhausted (rx/listener.go:214-217)                  */
#define SDR_ERR_STATE 7      /* call not valid in the current state                               */
#define SDR_ERR_WOULD_BLOCK 8 /* sdr_poll: no finished batch is waiting to be delivered                */

/* rx/receiver.go:15-27 */
#define SDR_CUMULATION_SIZE 100
#define SDR_NOISE_WINDOW 60
#define SDR_DBM_SHIFT 120
#define SDR_DEFAULT_PEAK_THRESHOLD 15.0f
#define SDR_DEFAULT_EDGE_WIDTH 70
#define SDR_DEFAULT_LISTENER_POOL_SIZE 30

typedef struct sdr_bank sdr_bank;

typedef struct sdr_config {
    int32_t struct_size;      /* = sizeof(sdr_config), ABI guard                                  */
    int32_t n_bands;          /* independent receivers in this bank                               */
    int32_t sample_rate;      /* Receiver.Start(sampleRate, blockSize)                            */
    int32_t block_size;       /* complex samples per frame; power of two in [512, 16384]          */
    int32_t edge_width;       /* bins ignored at both spectrum edges (default 70)                 */
    float peak_threshold;     /* dB over the noise floor for the peak scan (default 15)           */
    int32_t signal_debounce;  /* BoolDebouncer threshold of new listeners (default 1)             */
    int32_t max_listeners;    /* listener pool size per band (reference: 30 strain / 1 decode)    */
    int32_t max_batch_frames; /* capacity: frames per band per process call                       */
    int32_t max_peaks;        /* capacity: peaks reported per completed cumulation                */
    int32_t find_peaks;       /* 1: run FindPeaks on every completed 100-frame cumulation         */
    int32_t trace;            /* 1: keep per-frame value / raw / debounced traces (parity, scope) */
    int32_t device_id;        /* HIP device ordinal                                               */
    int32_t reserved;
} sdr_config;

/* dsp.Peak[float32,int] (dsp/fft.go:179-188), fixed-width */
typedef struct sdr_peak {
    int32_t from, to;
    int64_t from_frequency, to_frequency, signal_frequency;
    float signal_value;
    int32_t signal_bin;
} sdr_peak;

/* per-frame locals of Receiver.run (rx/receiver.go:381-385,394) */
typedef struct sdr_frame_rec {
    float min_mean;    /* FindNoiseFloor: T(minValue)                      */
    float dev_in;      /* value put into noiseDeviationMean                */
    double variance;   /* FindNoiseFloor: variance (consumed only through dev_in: see sdr_read_frame_records) */
    float nf_in;       /* value put into noiseFloorMean                    */
    float noise_dev;   /* noiseDeviation                                   */
    float noise_floor; /* noiseFloor                                       */
    float peak_thr;    /* peakThreshold = r.peakThreshold + noiseFloor     */
    float listen_thr;  /* noiseFloor + noiseDeviation (Listener.Listen)    */
    float pad;
} sdr_frame_rec;

/* one keying edge of a listener: debounced state changed at this frame (counted from bank start) */
typedef struct sdr_edge {
    uint32_t frame;
    uint32_t state; /* 1 = key down */
} sdr_edge;

const char *sdr_last_error(void);
int sdr_abi_version(void);

/* lifecycle ---------------------------------------------------------------------------------- */
int sdr_create(const sdr_config *cfg, sdr_bank **out);
int sdr_destroy(sdr_bank *bank);
/* Checks, on device `device_id`, the one piece of UNDOCUMENTED hardware behaviour a code path of the library depends on:
 * that the float64 matrix instruction the ORDERED variance chains of FindNoiseFloor can run on (dsp/fft.go:244-249: `sum +=
 * term`, one rounding per step; csrc/k_noise.hip) adds its four terms one after the other, each step rounded, in order -
 * 1024 wide-range quadruples against the same chains on the vector ALU, bit for bit.  0 = as assumed.  Since round 5 the
 * default noise-floor path (csrc/k_noise_scan.hip: values certified where they are consumed, literal loops on the vector
 * ALU for the rest) does not use that instruction, and sdr_create runs the check - once per device and process, refusing
 * to create a bank (SDR_ERR_HIP, message in sdr_last_error) where it fails - only when the environment selects the
 * ordered chains (SDR_NOISE_PATH=chains).  A host has no need to call it; one that does should do so AFTER its first
 * sdr_create: the probe launches a kernel, and HIP deals its four hardware queues to streams as they come - work issued
 * before a bank's streams exist can leave two of them sharing a queue (measured: graph mode at two thirds of its rate).
 * The caller's current device is left as it was. */
int sdr_self_check(int device_id);
/* Run on this hipStream_t (NULL = the null stream).  Must be called before the first process call
 * or while the bank is idle. */
int sdr_set_stream(sdr_bank *bank, void *hip_stream);

/* producer side ------------------------------------------------------------------------------ */
/* Copies n_floats/(2*block_size) interleaved I,Q float32 frames of `band` from host memory into the
 * bank's pinned staging queue (the input is borrowed only for the duration of the call). */
int sdr_push_iq(sdr_bank *bank, int band, int sample_rate, const float *iq, size_t n_floats);
/* KiwiSDR source: `payload` is one "SND" websocket message body (kiwi/client.go:284-308): a 17-byte
 * header (flags, sequence number, S-meter, GPS) followed by big-endian int16 I,Q pairs.  The raw bytes
 * are staged and unpacked ON THE DEVICE to float32 = float32(int16) / 32767 when the batch is processed;
 * like kiwi.Process.IQData (kiwi/kiwi.go:94-105) the message must hold whole frames (BAD_SIZE
 * otherwise).  A band's batch must not mix this with sdr_push_iq (SDR_ERR_STATE). */
int sdr_push_kiwi_snd(sdr_bank *bank, int band, int sample_rate, const uint8_t *payload, size_t n_bytes);
/* Frames currently staged for `band`. */
int sdr_staged_frames(sdr_bank *bank, int band);
/* Uploads and processes min-over-bands staged frames; *n_frames_out = frames consumed per band. */
int sdr_process_staged(sdr_bank *bank, int *n_frames_out);
/* Same, but at most max_frames per band (lets a host stop at a cumulation boundary, where the
 * reference attaches a new listener: rx/receiver.go:409-426). */
int sdr_process_staged_limit(sdr_bank *bank, int max_frames, int *n_frames_out);
/* Processes n_frames per band of IQ already in device memory, layout [band][frame][block_size][2]
 * float32 (band stride = n_frames*2*block_size floats), 16-byte aligned (SDR_ERR_BAD_ARG otherwise).
 * Asynchronous on the bank's stream. */
int sdr_process_device(sdr_bank *bank, const float *iq_dev, int n_frames);
/* Blocks until everything queued on the bank's stream has finished. */
int sdr_sync(sdr_bank *bank);

/* per-signal entry --------------------------------------------------------------------------- */
/* Binds a fresh listener (new debouncer + new decoder, Reset) to spectrum bin `bin` of `band`,
 * effective from the next processed frame. */
int sdr_attach(sdr_bank *bank, int band, int bin, int *listener_id);
int sdr_detach(sdr_bank *bank, int band, int listener_id);
int sdr_listener_count(sdr_bank *bank, int band);
/* Flush the listener's pending character (cw.Decoder.stop, cw/decode.go:352-354). */
int sdr_listener_stop(sdr_bank *bank, int band, int listener_id);

/* control ------------------------------------------------------------------------------------ */
int sdr_set_peak_threshold(sdr_bank *bank, int band, float threshold);
int sdr_set_edge_width(sdr_bank *bank, int edge_width);
int sdr_set_signal_debounce(sdr_bank *bank, int band, int debounce);
int sdr_set_center_frequency(sdr_bank *bank, int band, int64_t frequency);
int sdr_set_find_peaks(sdr_bank *bank, int on);

/* consumer side (all synchronise with the stream first) --------------------------------------- */
/* Frames per band consumed by the last process call / since bank creation. */
int sdr_last_batch_frames(sdr_bank *bank);
int64_t sdr_total_frames(sdr_bank *bank);
/* Number of 100-frame cumulations completed by the last process call. */
int sdr_last_batch_chunks(sdr_bank *bank);
/* Peaks of completed cumulation `chunk` (0-based within the last batch).  *frame_in_batch = index
 * of the frame that completed it. */
int sdr_read_peaks(sdr_bank *bank, int band, int chunk, sdr_peak *out, int max, int *n_out, int *frame_in_batch);
/* Cumulated spectrum (sum over 100 frames, float32[block_size]) of that chunk (rx/receiver.go:404-407), every bin the
 * reference's ordered float32 sum.  The pipeline itself keeps a cumulation exact only where FindPeaks reads it; this call
 * recomputes the whole row from the batch's retained spectra and the carry it started from. */
int sdr_read_cumulation(sdr_bank *bank, int band, int chunk, float *out);
/* Decoded text of a listener since the last read, UTF-8 (what the reference writes to io.Writer). */
int sdr_read_text(sdr_bank *bank, int band, int listener_id, char *out, int max_bytes, int *n_bytes);
/* Keying edges of a listener produced by the last process call. */
int sdr_read_edges(sdr_bank *bank, int band, int listener_id, sdr_edge *out, int max, int *n_out);
/* Packed debounced on/off bits of the last batch: bit (f & 63) of word (f >> 6). */
int sdr_read_keying_bits(sdr_bank *bank, int band, int listener_id, uint64_t *out, int max_words);
/* The last batch's frame records.  min_mean, dev_in, nf_in and everything behind them are the reference's bits on the hot
 * path already (certified from order-free sums, or the literal loops: csrc/noise_cert.h); `variance`, which nothing
 * downstream reads as a float64, is recomputed here by the reference's ordered loop, and the certified fields are checked
 * against the literal ones on the way (SDR_ERR_STATE on a difference: a defect, never seen). */
int sdr_read_frame_records(sdr_bank *bank, int band, sdr_frame_rec *out, int max);
/* trace == 1 only: per-frame value handed to Listen, raw and debounced state of a listener. */
int sdr_read_trace(sdr_bank *bank, int band, int listener_id, float *values, uint8_t *raw, uint8_t *debounced, int max);
/* spectrum (dB+120, fftshifted) and psd of frame `frame` of the last batch, float32[block_size] each. */
int sdr_read_spectrum(sdr_bank *bank, int band, int frame, float *spectrum, float *psd);
/* cw.Decoder state of a listener: ticks, onStart, offStart, wpm, on{low,high,last,thr}, off{...}. */
int sdr_read_decoder_state(sdr_bank *bank, int band, int listener_id, double *out12);

/* bulk delivery ----------------------------------------------------------------------------- */
/* With results enabled every process call ends with two small kernels that copy what the batch produced - peaks
 * of each completed cumulation, each listener's keying edges and newly decoded runes - into pinned host memory;
 * sdr_poll hands the oldest finished batch to the caller WITHOUT draining the pipeline (it looks at two events)
 * and never loses one: a batch whose buffers are about to be reused is parked on the host first.  Text is then
 * delivered here only (sdr_read_text finds nothing left).  Decoded runes the device could not store and edges
 * beyond a batch's edge buffer are counted, never silently lost: runes_dropped / edges_dropped (both stay 0
 * while every batch is polled; the reference's io.Writer never drops). */
typedef struct sdr_chunk_result {
    int32_t band;
    int32_t n_peaks;    /* peaks[first_peak .. first_peak + n_peaks) */
    int64_t frame;      /* frame (counted from bank start) that completed this 100-frame cumulation */
    int32_t first_peak;
    int32_t peaks_found; /* runs FindPeaks found; > n_peaks only if max_peaks was too small */
} sdr_chunk_result;

typedef struct sdr_listener_result {
    int32_t band, listener;
    int32_t first_edge, n_edges; /* edges[first_edge ..): frame counted from bank start, state 1 = key down */
    int32_t first_rune, n_runes; /* runes[first_rune ..): Unicode code points in decode order */
} sdr_listener_result;

typedef struct sdr_results {
    int32_t struct_size;   /* = sizeof(sdr_results), ABI guard (in) */
    int32_t n_frames;      /* frames per band in this batch (out) */
    int64_t batch_index;   /* 0-based count of process calls (out) */
    int64_t first_frame;   /* bank frame index of the batch's first frame (out) */
    /* caller-owned buffers with their capacities in records (in); counts written (out).  If a buffer is too
     * small the call returns SDR_ERR_BAD_SIZE with the n_* fields set to what is needed and delivers nothing. */
    sdr_chunk_result *chunks;
    int32_t chunks_cap, n_chunks;
    sdr_peak *peaks;
    int32_t peaks_cap, n_peaks;
    sdr_listener_result *listeners; /* listeners with at least one edge or rune in this batch */
    int32_t listeners_cap, n_listeners;
    sdr_edge *edges;
    int32_t edges_cap, n_edges;
    uint32_t *runes;       /* Unicode code points */
    uint32_t *rune_frames; /* same capacity and count as runes: bank frame index of the Tick that wrote each rune */
    int32_t runes_cap, n_runes;
    uint64_t runes_dropped, edges_dropped; /* since bank creation (out) */
} sdr_results;

int sdr_enable_results(sdr_bank *bank, int on);
/* Oldest finished, undelivered batch -> *r.  SDR_ERR_WOULD_BLOCK if there is none (yet); with wait != 0 the
 * call blocks until the oldest undelivered batch has finished (WOULD_BLOCK only if nothing was processed). */
int sdr_poll(sdr_bank *bank, sdr_results *r, int wait);

/* Strain-mode discovery over a long batch (rx/receiver.go:404-426: one listener bound per completed cumulation, to a
 * peak of that cumulation, listening from the very next frame).  With deferral on (needs sdr_enable_results), a
 * sdr_process_* call runs the spectral half of the batch only - FFT, noise floor, thresholds, cumulations and FindPeaks
 * of EVERY cumulation the batch completes - and the bank refuses further batches until sdr_process_listen has run the
 * listeners over the retained spectra.  In between the host reads the peaks (sdr_poll_peaks: the chunks / peaks part of
 * sdr_results, the batch stays undelivered; sdr_poll delivers it whole once the listen half has run) and binds listeners
 * with sdr_attach_at: like sdr_attach, but the listener listens from bank frame `start_frame` on (counted from the
 * bank's first frame; between the first frame of the waiting batch and the next frame to be processed).  Frames before
 * start_frame never reach its debouncer or decoder, so it produces exactly what a listener attached there frame by frame
 * would have.  Neither call waits for the device. */
int sdr_defer_listen(sdr_bank *bank, int on);
int sdr_listen_pending(sdr_bank *bank);
int sdr_poll_peaks(sdr_bank *bank, sdr_results *results, int wait);
int sdr_attach_at(sdr_bank *bank, int band, int bin, int64_t start_frame, int *listener_id);
int sdr_process_listen(sdr_bank *bank);
/* Batches processed but not yet delivered. */
int sdr_results_pending(sdr_bank *bank);
/* Overflow counters without bulk delivery (synchronises): runes the decoders could not store because a text
 * buffer was full, and keying edges beyond the edge buffer of the batches so far. */
int sdr_read_drop_counters(sdr_bank *bank, uint64_t *runes_dropped, uint64_t *edges_dropped);

/* graph mode --------------------------------------------------------------------------------- */
/* The steady state of Receiver.run (rx/receiver.go:353-463) for sdr_graph_batches() consecutive batches of
 * n_frames frames each, recorded once and replayed: one kernel-only hipGraph per stream of the bank (two on the peaks
 * stream), ordered inside a replay by events around whole graphs, with up to four replays in flight over buffer sets of
 * their own (allocated at capture), so that consecutive replays overlap stage by stage like consecutive eager batches.
 * Needs a bank on a real stream (sdr_set_stream with a non-null stream) and, once captured, all processing to go
 * through sdr_graph_launch (sdr_process_* return SDR_ERR_STATE until sdr_graph_release).  Attaching or detaching a
 * listener, sdr_enable_results and sdr_set_find_peaks invalidate the capture (sdr_graph_launch returns SDR_ERR_STATE:
 * capture again).  Results are read / polled exactly as after sdr_process_device;
 * the "last batch" of the read calls is the last replay's last. */
int sdr_graph_batches(sdr_bank *bank);
int sdr_graph_capture(sdr_bank *bank, int n_frames);
/* iq_dev: sdr_graph_batches() device pointers, one batch each, layout and alignment as sdr_process_device. */
int sdr_graph_launch(sdr_bank *bank, const float *const *iq_dev);
/* Back to sdr_process_*: drains the pipeline, moves what was not polled yet to the host-side queue (sdr_poll keeps
 * delivering it, oldest first) and frees the replays' buffer sets; what the last replay's last batch left on the device
 * goes with them - read it (sdr_read_*) before the release. */
int sdr_graph_release(sdr_bank *bank);

/* scope tap ---------------------------------------------------------------------------------- */
/* The reference shows its inner workings through scope.Scope (scope/scope.go:33-37); NullScope is the default
 * and so is "no tap" here: the two reads below need a bank created with trace = 1 (that is scope.Active()).
 * A frame's Timestamp is replaced by the frame index, everything else is the reference's payload:
 *   stream "spectrum" (rx/receiver.go:428-457): once per completed cumulation, Values = cumulation / 100 as
 *       float64, FrequencyMarkers{"signal_bin"} = bin of the first listener (-1: none),
 *       MagnitudeMarkers{"threshold"} = peakThreshold, FromFrequency 0, ToFrequency 1;
 *   stream "demod" (cw/spectral.go:56-81): once per listener per frame, Values{"threshold", "value",
 *       "state" (-1 / 100), "debounced" (-1 / 80)}. */
typedef struct sdr_scope_spectral_frame {
    int64_t frame;          /* bank frame index that completed the cumulation */
    double from_frequency;  /* 0 */
    double to_frequency;    /* 1 */
    double signal_bin;      /* FrequencyMarkers["signal_bin"] */
    double threshold;       /* MagnitudeMarkers["threshold"] */
    int32_t n_values;       /* block_size */
    int32_t reserved;
} sdr_scope_spectral_frame;

typedef struct sdr_scope_time_frame {
    double threshold, value, state, debounced;
} sdr_scope_time_frame;

/* 1 if the bank taps (created with trace = 1), else 0: scope.Scope.Active(). */
int sdr_scope_active(sdr_bank *bank);
/* Spectral frame of completed cumulation `chunk` of the last batch; values: float64[block_size]. */
int sdr_scope_read_spectral(sdr_bank *bank, int band, int chunk, sdr_scope_spectral_frame *frame, double *values, int max_values);
/* The listener's "demod" time frames of the last batch, one per frame. */
int sdr_scope_read_demod(sdr_bank *bank, int band, int listener_id, sdr_scope_time_frame *out, int max, int *n_out);
/* cw.Decoder's scope streams (cw/decode.go:228-243, :433-491) for the last batch, one record per tick the listener
 * took: scopeDecode {duration, on_threshold, state}, scopeSignalTiming {on_duration = state ? duration : 0, on_threshold,
 * _low, _high, 2 x _high, state}, scopeGapTiming {off_duration = state ? 0 : duration, off_threshold, _low, _high,
 * 2 x _high - threshold, state}, scopeSignal {state}: every channel is one of these fields or a sum of two.  *n_out: the
 * number of ticks (a listener bound inside the batch has fewer than the batch has frames).  trace == 1 only. */
typedef struct sdr_scope_decode_frame {
    int64_t frame;     /* bank frame index of the tick */
    double duration;   /* currentDuration: ticks since the last edge */
    double state;      /* 0 / 1 */
    double on_threshold, on_threshold_low, on_threshold_high;
    double off_threshold, off_threshold_low, off_threshold_high;
} sdr_scope_decode_frame;
int sdr_scope_read_decode(sdr_bank *bank, int band, int listener_id, sdr_scope_decode_frame *out, int max, int *n_out);

/* measurement ------------------------------------------------------------------------------- */
/* When enabled every kernel launch is bracketed by HIP events on the bank's stream. */
int sdr_profile_enable(sdr_bank *bank, int on);
/* kernel: 0 fft_project, 1 window_means, 2 noise_stats, 3 thresholds, 4 listen_gather,
 *         5 cumulate, 6 find_peaks, 7 listen_decode.  Returns accumulated milliseconds and launch count. */
int sdr_profile_read(sdr_bank *bank, int kernel, double *total_ms, int *launches);
int sdr_profile_reset(sdr_bank *bank);
const char *sdr_kernel_name(int kernel);

/* audio path (cw/audio.go): n_streams mono float32 streams, one Goertzel + debouncer + decoder each */
typedef struct sdr_audio sdr_audio;
int sdr_audio_create(int n_streams, double pitch, int sample_rate, int max_blocks, int device_id, sdr_audio **out);
int sdr_audio_destroy(sdr_audio *a);
int sdr_audio_blocksize(sdr_audio *a);
int sdr_audio_set_scale(sdr_audio *a, double scale);             /* 0 = autoscale (audio.go:184-187) */
int sdr_audio_set_debounce(sdr_audio *a, int threshold);          /* default 3 (audio.go:18)          */
int sdr_audio_set_magnitude_threshold(sdr_audio *a, double t);    /* default 0.75 (dsp.go:12)         */
/* Feed n_samples mono float32 samples per stream (host memory, layout [stream][n_samples]); whole
 * Goertzel blocks are processed, the remainder is kept for the next call (audio.go:175-179). */
int sdr_audio_write(sdr_audio *a, const float *samples, int n_samples);
int sdr_audio_close(sdr_audio *a);                                /* decoder.stop (audio.go:205-207)  */
int sdr_audio_read_text(sdr_audio *a, int stream, char *out, int max_bytes, int *n_bytes);
/* per-block normalised magnitude / raw / debounced state of the last write (parity) */
int sdr_audio_read_trace(sdr_audio *a, int stream, double *magnitudes, uint8_t *raw, uint8_t *debounced, int max,
                         int *n_blocks);

#ifdef __cplusplus
}
#endif
#endif /* SDRAINER_HIP_H */
