// capi_read.hip — per-batch / per-listener reads of what the last batch left on the device (tests, one-off reads: they
// drain the pipeline) and the scope tap (scope/scope.go:14-37).  Part of the C ABI, see bank.h.
#include <string>

#include "bank.h"
#include "twiddles.h"

using namespace sdrcapi;

namespace {
size_t utf8_encode(uint32_t r, char *out)
{
    if (r < 0x80) {
        out[0] = (char)r;
        return 1;
    }
    if (r < 0x800) {
        out[0] = (char)(0xC0 | (r >> 6));
        out[1] = (char)(0x80 | (r & 0x3F));
        return 2;
    }
    out[0] = (char)(0xE0 | (r >> 12));
    out[1] = (char)(0x80 | ((r >> 6) & 0x3F));
    out[2] = (char)(0x80 | (r & 0x3F));
    return 3;
}

// Completed cumulation `chunk` of the last batch, exact in every bin.  The pipeline keeps a cumulation exact only where
// FindPeaks reads it (k_peaks.hip: an upper bound elsewhere); a reader gets the whole row recomputed from the batch's
// retained psd rows and the carry it started from - the same kernel that carries an open cumulation across batches.
// (pipeline drained by the caller)
int exact_cumulation_row(sdr_bank *b, int band, int chunk, float *out)
{
    const sdr_config &c = b->cfg;
    const size_t N = (size_t)c.block_size;
    const BatchSet &S = b->set[b->last_set];
    HIP_TRY(hipSetDevice(b->device));
    const sdr::CumGeom cg{c.block_size, c.max_batch_frames, b->last_frames, b->last_count0, b->max_chunks};
    HIP_TRY(sdr::launch_cumulation_row(S.psd.p + (size_t)band * c.max_batch_frames * N, b->db_tab.p, b->carry[b->last_carry_in].p + (size_t)band * N,
                                       b->spectrum_row.p, cg, chunk, b->stream[S_FFT]));
    HIP_TRY(hipStreamSynchronize(b->stream[S_FFT]));
    HIP_TRY(hipMemcpy(out, b->spectrum_row.p, sizeof(float) * N, hipMemcpyDeviceToHost));
    return SDR_OK;
}
}  // namespace

extern "C" {
#pragma GCC visibility push(default)

int sdr_read_peaks(sdr_bank *b, int band, int chunk, sdr_peak *out, int max, int *n_out, int *frame_in_batch)
{
    int rc = check_band(b, band);
    if (rc)
        return rc;
    if (chunk < 0 || chunk >= b->last_chunks)
        return fail(SDR_ERR_BAD_ARG, "chunk out of range");
    rc = sync_bank(b);
    if (rc)
        return rc;
    const sdr_config &c = b->cfg;
    const BatchSet &S = b->set[b->last_set];
    if (frame_in_batch)
        *frame_in_batch = (SDR_CUMULATION_SIZE - b->last_count0) + chunk * SDR_CUMULATION_SIZE - 1;
    if (!b->find_peaks) {
        if (n_out)
            *n_out = 0;
        return SDR_OK;
    }
    int count = 0;
    HIP_TRY(hipMemcpy(&count, S.peak_counts.p + (size_t)band * b->max_chunks + chunk, sizeof(int), hipMemcpyDeviceToHost));
    if (n_out)
        *n_out = count;
    const int n = std::min(std::min(count, c.max_peaks), max);
    if (n <= 0 || !out)
        return SDR_OK;
    std::vector<sdr::DevPeak> dp((size_t)n);
    HIP_TRY(hipMemcpy(dp.data(), S.dev_peaks.p + ((size_t)band * b->max_chunks + chunk) * c.max_peaks,
                      sizeof(sdr::DevPeak) * (size_t)n, hipMemcpyDeviceToHost));
    host::FrequencyMapping fm(c.sample_rate, c.block_size, b->center_frequency[band]);
    for (int i = 0; i < n; i++) {
        const sdr::DevPeak &p = dp[i];
        sdr_peak &o = out[i];
        o.from = p.from;
        o.to = p.to;
        o.signal_bin = p.signal_bin;
        o.signal_value = p.signal_value;
        o.from_frequency = fm.BinToFrequency(p.from, host::BinFrom);
        o.to_frequency = fm.BinToFrequency(p.to, host::BinTo);
        const double corr = host::PeakCenterCorrection(p.signal_bin, c.block_size, p.y1, p.y2, p.y3);
        o.signal_frequency = fm.BinToFrequency(p.signal_bin, corr);
    }
    return SDR_OK;
}

int sdr_read_cumulation(sdr_bank *b, int band, int chunk, float *out)
{
    int rc = check_band(b, band);
    if (rc)
        return rc;
    if (chunk < 0 || chunk >= b->last_chunks || !out)
        return fail(SDR_ERR_BAD_ARG, "chunk out of range");
    rc = sync_bank(b);
    if (rc)
        return rc;
    // (tests only, SDR_READ_CUM_RAW=1: the row as the pipeline keeps it - an upper bound outside the bins FindPeaks reads -
    // so that a test can check bound >= exact in every bin of every cumulation: tests/test_gpu_parity_bench_sizes.py)
    if (const char *raw = getenv("SDR_READ_CUM_RAW"))
        if (raw[0] == '1') {
            HIP_TRY(hipMemcpy(out, b->set[b->last_set].cum_out.p + ((size_t)band * b->max_chunks + chunk) * b->cfg.block_size,
                              sizeof(float) * (size_t)b->cfg.block_size, hipMemcpyDeviceToHost));
            return SDR_OK;
        }
    return exact_cumulation_row(b, band, chunk, out);
}

int sdr_read_text(sdr_bank *b, int band, int lid, char *out, int max_bytes, int *n_bytes)
{
    int rc = check_listener(b, band, lid);
    if (rc)
        return rc;
    rc = sync_bank(b);
    if (rc)
        return rc;
    const size_t idx = (size_t)band * b->cfg.max_listeners + lid;
    sdr::ListenerSlot s;
    HIP_TRY(hipMemcpy(&s, b->slots.p + idx, sizeof s, hipMemcpyDeviceToHost));
    std::vector<uint32_t> runes(s.text_count);
    if (s.text_count)
        HIP_TRY(hipMemcpy(runes.data(), b->text.p + idx * b->text_cap, sizeof(uint32_t) * s.text_count,
                          hipMemcpyDeviceToHost));
    int n = 0;
    uint32_t consumed = 0;
    for (; consumed < s.text_count; consumed++) {
        char tmp[4];
        const size_t k = utf8_encode(runes[consumed], tmp);
        if (n + (int)k > max_bytes)
            break;
        if (out)
            memcpy(out + n, tmp, k);
        n += (int)k;
    }
    if (n_bytes)
        *n_bytes = n;
    // drop what was handed out, keep the rest at the front of the buffer
    const uint32_t left = s.text_count - consumed;
    if (left && consumed)
        HIP_TRY(hipMemcpy(b->text.p + idx * b->text_cap, runes.data() + consumed, sizeof(uint32_t) * left,
                          hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(&b->slots.p[idx].text_count, &left, sizeof(uint32_t), hipMemcpyHostToDevice));
    return SDR_OK;
}

int sdr_read_edges(sdr_bank *b, int band, int lid, sdr_edge *out, int max, int *n_out)
{
    int rc = check_listener(b, band, lid);
    if (rc)
        return rc;
    rc = sync_bank(b);
    if (rc)
        return rc;
    const size_t idx = (size_t)band * b->cfg.max_listeners + lid;
    const BatchSet &S = b->set[b->last_set];
    uint32_t count = 0;
    HIP_TRY(hipMemcpy(&count, S.edge_counts.p + idx, sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (n_out)
        *n_out = (int)count;
    const int n = std::min(std::min((int)count, b->edge_cap), max);
    if (n > 0 && out)
        HIP_TRY(hipMemcpy(out, S.edges.p + idx * b->edge_cap, sizeof(sdr_edge) * (size_t)n, hipMemcpyDeviceToHost));
    return SDR_OK;
}

int sdr_read_keying_bits(sdr_bank *b, int band, int lid, uint64_t *out, int max_words)
{
    int rc = check_listener(b, band, lid);
    if (rc)
        return rc;
    rc = sync_bank(b);
    if (rc)
        return rc;
    const size_t idx = (size_t)band * b->cfg.max_listeners + lid;
    const int words = std::min((b->last_frames + 63) / 64, max_words);
    if (words > 0 && out)
        HIP_TRY(hipMemcpy(out, b->set[b->last_set].bits.p + idx * b->bit_words, sizeof(uint64_t) * (size_t)words,
                          hipMemcpyDeviceToHost));
    return SDR_OK;
}

int sdr_read_frame_records(sdr_bank *b, int band, sdr_frame_rec *out, int max)
{
    int rc = check_band(b, band);
    if (rc)
        return rc;
    rc = sync_bank(b);
    if (rc)
        return rc;
    const int n = std::min(b->last_frames, max);
    if (n > 0 && out) {
        BatchSet &S = b->set[b->last_set];
        sdr_frame_rec *recs = S.recs.p + (size_t)band * b->cfg.max_batch_frames;
        // The hot path (k_noise_scan.hip) produces FindNoiseFloor's CONSUMED values and only brackets the float64 variance,
        // which nothing consumes.  A read gets the reference's variance - the literal loops run here, for every frame of
        // the batch - and the same pass compares the consumed values with the literal ones: a difference is an error.
        static const bool scan_path = !(getenv("SDR_NOISE_PATH") && std::string(getenv("SDR_NOISE_PATH")) == "chains");
        if (scan_path) {
            unsigned *mism = nullptr, h = 0;
            HIP_TRY(hipMalloc(&mism, sizeof(unsigned)));
            hipError_t e = hipMemset(mism, 0, sizeof(unsigned));
            if (e == hipSuccess)
                e = sdr::launch_noise_exact_check(S.psd.p + (size_t)band * b->cfg.max_batch_frames * b->cfg.block_size, recs, b->noise_geom(),
                                                  b->last_frames, mism, nullptr);
            if (e == hipSuccess)
                e = hipMemcpy(&h, mism, sizeof h, hipMemcpyDeviceToHost);
            (void)hipFree(mism);
            if (e != hipSuccess)
                return fail(SDR_ERR_HIP, std::string("exact FindNoiseFloor: ") + hipGetErrorString(e));
            if (h)
                return fail(SDR_ERR_HIP, std::to_string(h) + " frame(s) whose certified FindNoiseFloor values differ from the literal algorithm's");
        }
        HIP_TRY(hipMemcpy(out, recs, sizeof(sdr_frame_rec) * (size_t)n, hipMemcpyDeviceToHost));
    }
    return SDR_OK;
}

int sdr_read_trace(sdr_bank *b, int band, int lid, float *values, uint8_t *raw, uint8_t *debounced, int max)
{
    int rc = check_listener(b, band, lid);
    if (rc)
        return rc;
    if (!b->cfg.trace)
        return fail(SDR_ERR_STATE, "bank was created without trace");
    rc = sync_bank(b);
    if (rc)
        return rc;
    const sdr_config &c = b->cfg;
    const BatchSet &S = b->set[b->last_set];
    const int n = std::min(b->last_frames, max);
    if (n <= 0)
        return SDR_OK;
    const size_t L = (size_t)c.max_listeners;
    const size_t base = (size_t)band * c.max_batch_frames * L + lid;
    // strided gather: [frame][listener] -> per-listener row
    if (values)
        HIP_TRY(hipMemcpy2D(values, sizeof(float), S.tr_values.p + base, sizeof(float) * L, sizeof(float), (size_t)n,
                            hipMemcpyDeviceToHost));
    if (raw)
        HIP_TRY(hipMemcpy2D(raw, 1, S.tr_raw.p + base, L, 1, (size_t)n, hipMemcpyDeviceToHost));
    if (debounced)
        HIP_TRY(hipMemcpy2D(debounced, 1, S.tr_deb.p + base, L, 1, (size_t)n, hipMemcpyDeviceToHost));
    return SDR_OK;
}

int sdr_read_spectrum(sdr_bank *b, int band, int frame, float *spectrum, float *psd)
{
    int rc = check_band(b, band);
    if (rc)
        return rc;
    if (frame < 0 || frame >= b->last_frames)
        return fail(SDR_ERR_BAD_ARG, "frame out of range");
    rc = sync_bank(b);
    if (rc)
        return rc;
    const size_t N = (size_t)b->cfg.block_size;
    const size_t off = ((size_t)band * b->cfg.max_batch_frames + frame) * N;
    const BatchSet &S = b->set[b->last_set];
    if (spectrum) {
        // the pipeline keeps psd only; the dB projection of the row is made on demand (dsp/fft.go:79-81)
        HIP_TRY(sdr::launch_spectrum_row(S.psd.p + off, b->spectrum_row.p, (int)N, b->stream[S_FFT]));
        HIP_TRY(hipStreamSynchronize(b->stream[S_FFT]));
        HIP_TRY(hipMemcpy(spectrum, b->spectrum_row.p, sizeof(float) * N, hipMemcpyDeviceToHost));
    }
    if (psd)
        HIP_TRY(hipMemcpy(psd, S.psd.p + off, sizeof(float) * N, hipMemcpyDeviceToHost));
    return SDR_OK;
}

int sdr_read_decoder_state(sdr_bank *b, int band, int lid, double *out12)
{
    int rc = check_listener(b, band, lid);
    if (rc)
        return rc;
    if (!out12)
        return fail(SDR_ERR_BAD_ARG, "null out");
    rc = sync_bank(b);
    if (rc)
        return rc;
    sdr::ListenerSlot s;
    HIP_TRY(hipMemcpy(&s, b->slots.p + (size_t)band * b->cfg.max_listeners + lid, sizeof s, hipMemcpyDeviceToHost));
    const cw::DecoderState &d = s.dec;
    out12[0] = d.ticks;
    out12[1] = d.onStart;
    out12[2] = d.offStart;
    out12[3] = d.wpm;
    out12[4] = d.onThreshold.low;
    out12[5] = d.onThreshold.high;
    out12[6] = d.onThreshold.last;
    out12[7] = d.onThreshold.threshold;
    out12[8] = d.offThreshold.low;
    out12[9] = d.offThreshold.high;
    out12[10] = d.offThreshold.last;
    out12[11] = d.offThreshold.threshold;
    return SDR_OK;
}

// ---- scope tap (scope/scope.go:14-37) -------------------------------------------------------------------------
int sdr_scope_active(sdr_bank *b) { return b ? (b->cfg.trace ? 1 : 0) : 0; }

int sdr_scope_read_spectral(sdr_bank *b, int band, int chunk, sdr_scope_spectral_frame *frame, double *values, int max_values)
{
    int rc = check_band(b, band);
    if (rc)
        return rc;
    if (!b->cfg.trace)
        return fail(SDR_ERR_STATE, "scope inactive: the bank was created without trace");
    if (chunk < 0 || chunk >= b->last_chunks || !frame)
        return fail(SDR_ERR_BAD_ARG, "chunk out of range");
    rc = sync_bank(b);
    if (rc)
        return rc;
    const sdr_config &c = b->cfg;
    const int N = c.block_size;
    const BatchSet &S = b->set[b->last_set];
    const int end_frame = (SDR_CUMULATION_SIZE - b->last_count0) + chunk * SDR_CUMULATION_SIZE - 1;
    sdr_frame_rec rec;
    HIP_TRY(hipMemcpy(&rec, S.recs.p + (size_t)band * c.max_batch_frames + end_frame, sizeof rec, hipMemcpyDeviceToHost));
    frame->frame = (b->total_frames - b->last_frames) + end_frame;
    frame->from_frequency = 0.0;
    frame->to_frequency = 1.0;
    frame->threshold = (double)rec.peak_thr;
    frame->n_values = N;
    frame->reserved = 0;
    // DecodeMode: the listener; StrainMode: the pool's first listener (rx/receiver.go:430-441); -1 without one
    frame->signal_bin = -1.0;
    for (int i = 0; i < b->n_slots[band]; i++) {
        const sdr::ListenerSlot &sl = b->h_slots[(size_t)band * c.max_listeners + i];
        if (sl.active) {
            frame->signal_bin = (double)sl.bin;
            break;
        }
    }
    if (values) {
        std::vector<float> cum((size_t)N);
        if ((rc = exact_cumulation_row(b, band, chunk, cum.data())))
            return rc;
        const double scale = 1.0 / (double)SDR_CUMULATION_SIZE;  // scaledValuesForScope(cumulation, 1.0/float64(cumulationSize))
        for (int i = 0; i < std::min(N, max_values); i++)
            values[i] = (double)cum[i] * scale;
    }
    return SDR_OK;
}

int sdr_scope_read_demod(sdr_bank *b, int band, int lid, sdr_scope_time_frame *out, int max, int *n_out)
{
    int rc = check_listener(b, band, lid);
    if (rc)
        return rc;
    if (!b->cfg.trace)
        return fail(SDR_ERR_STATE, "scope inactive: the bank was created without trace");
    const int n = std::min(b->last_frames, max);
    if (n_out)
        *n_out = b->last_frames;
    if (n <= 0 || !out)
        return SDR_OK;
    std::vector<float> v((size_t)n);
    std::vector<uint8_t> raw((size_t)n), deb((size_t)n);
    rc = sdr_read_trace(b, band, lid, v.data(), raw.data(), deb.data(), n);  // synchronises
    if (rc)
        return rc;
    std::vector<sdr_frame_rec> recs((size_t)n);
    HIP_TRY(hipMemcpy(recs.data(), b->set[b->last_set].recs.p + (size_t)band * b->cfg.max_batch_frames,
                      sizeof(sdr_frame_rec) * (size_t)n, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) {
        out[i].threshold = (double)recs[i].listen_thr;
        out[i].value = (double)v[i];
        out[i].state = raw[i] ? 100.0 : -1.0;      // cw/spectral.go:61-64
        out[i].debounced = deb[i] ? 80.0 : -1.0;   // cw/spectral.go:65-68
    }
    return SDR_OK;
}

// cw.Decoder's scope streams (cw/decode.go:228-243 and :433-491): per tick, what scopeDecode / scopeSignalTiming /
// scopeGapTiming / scopeSignal show - the current run's duration, both adaptive thresholds with their low and high, the
// state.  The decoders run on the device in closed form between edges and keep none of this per tick; it is replayed
// here, on the host, tick by tick with the literal Tick (cw_decoder.h decoder_tick, the function the device's run-length
// form is checked against), from the decoder's state before the batch and the batch's debounced keying.
int sdr_scope_read_decode(sdr_bank *b, int band, int lid, sdr_scope_decode_frame *out, int max, int *n_out)
{
    int rc = check_listener(b, band, lid);
    if (rc)
        return rc;
    if (!b->cfg.trace)
        return fail(SDR_ERR_STATE, "scope inactive: the bank was created without trace");
    if (n_out)
        *n_out = 0;
    const int frames = b->last_frames;
    if (frames <= 0)
        return SDR_OK;
    std::vector<uint8_t> deb((size_t)frames);
    rc = sdr_read_trace(b, band, lid, nullptr, nullptr, deb.data(), frames);  // synchronises
    if (rc)
        return rc;
    const BatchSet &S = b->set[b->last_set];
    sdr::ListenerSlot slot;
    HIP_TRY(hipMemcpy(&slot, S.slots_before.p + (size_t)band * b->cfg.max_listeners + lid, sizeof slot, hipMemcpyDeviceToHost));
    if (!slot.active)
        return fail(SDR_ERR_STATE, "listener was not attached during the last batch");
    // a listener bound inside the batch (sdr_attach_at) ticks from its first frame on
    const int64_t first = b->total_frames - frames;
    const int skip = (int)std::max<int64_t>(0, (int64_t)(int32_t)(slot.start_frame - (uint32_t)first));
    std::vector<uint16_t> table(cw::kMorseTableSize);
    cw::build_morse_table(table.data());
    struct NullSink {
        void put(uint32_t) {}
    } sink;
    cw::DecoderState d = slot.dec;
    int n = 0;
    for (int f = skip; f < frames; f++) {
        const bool state = deb[(size_t)f] != 0;
        cw::decoder_tick(d, state, table.data(), sink);
        if (out && n < max) {
            sdr_scope_decode_frame &o = out[n];
            o.frame = first + f;
            o.duration = state ? d.ticks - d.onStart : d.ticks - d.offStart;  // currentDuration :222-227
            o.state = state ? 1.0 : 0.0;
            o.on_threshold = d.onThreshold.threshold;
            o.on_threshold_low = d.onThreshold.low;
            o.on_threshold_high = d.onThreshold.high;
            o.off_threshold = d.offThreshold.threshold;
            o.off_threshold_low = d.offThreshold.low;
            o.off_threshold_high = d.offThreshold.high;
        }
        n++;
    }
    if (n_out)
        *n_out = n;
    return SDR_OK;
}

#pragma GCC visibility pop
}  // extern "C"
