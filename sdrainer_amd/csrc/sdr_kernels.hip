// sdr_kernels.hip — CDNA4 (gfx950) kernels of the IQ-strainer path.  Compiled with
// -ffp-contract=off: every float op below is meant to round exactly as the Go reference does.
//
// Per batch and band (F frames of N complex64 samples):
//   k_fft_project   1 workgroup / frame   IQ -> float64 radix-2 DIT FFT (fft_f64.h) -> fftshift ->
//                                          psd = f32(|X|^2), spectrum = f32(10 log10(20 psd/N^2)) + 120
//   k_window_means  1 lane / (frame,window)  FindNoiseFloor pass 1: sequential f64 window sums
//   k_noise_stats   1 lane / frame           FindNoiseFloor pass 2 (min window, quirky variance) + dB inputs
//   k_thresholds    1 workgroup / band       the two 60-frame RollingMeans in frame order -> thresholds
//   k_listen        1 wave / tracked signal  gather bin, compare, debounce, Morse state machine
//   k_cumulate      1 lane / 4 bins / chunk  ordered f32 cumulation over 100 frames
//   k_find_peaks    1 workgroup / chunk      run-length peak scan of a completed cumulation
#include <hip/hip_runtime.h>

#include "../../include/sdrainer_hip.h"
#include "cw_decoder.h"
#include "fft_f64.h"
#include "gomath.h"
#include "sdr_device.h"

namespace sdr {

// ---------------------------------------------------------------------------------------------
// k_fft_project  (dsp/fft.go:23-37 IQToSpectrumAndPSD + rx/receiver.go:376-378 projection closure)
// ---------------------------------------------------------------------------------------------
template <int LOGN, int P>
__device__ __forceinline__ void run_passes(double (&xr)[fft64::Plan<LOGN>::R], double (&xi)[fft64::Plan<LOGN>::R],
                                           int t, const fft64::cplx *__restrict__ tw, double *lds)
{
    using PL = fft64::Plan<LOGN>;
    fft64::butterfly_pass<LOGN, P>(xr, xi, t, tw);
    if constexpr (P < PL::NPASS - 1) {
        if constexpr (P > 0)
            __syncthreads();  // everyone is done reading the previous exchange
        if constexpr (PL::SPLIT) {
            fft64::exchange_write<LOGN, P>(xr, t, lds);
            __syncthreads();
            fft64::exchange_read<LOGN, P>(xr, t, lds);
            __syncthreads();
            fft64::exchange_write<LOGN, P>(xi, t, lds);
            __syncthreads();
            fft64::exchange_read<LOGN, P>(xi, t, lds);
        } else {
            fft64::exchange_write<LOGN, P>(xr, t, lds);
            fft64::exchange_write<LOGN, P>(xi, t, lds + PL::N);
            __syncthreads();
            fft64::exchange_read<LOGN, P>(xr, t, lds);
            fft64::exchange_read<LOGN, P>(xi, t, lds + PL::N);
        }
        run_passes<LOGN, P + 1>(xr, xi, t, tw, lds);
    }
}

template <int LOGN>
__global__ __launch_bounds__(fft64::Plan<LOGN>::T) void k_fft_project(const float *__restrict__ iq,
                                                                      const fft64::cplx *__restrict__ tw,
                                                                      float *__restrict__ spectrum,
                                                                      float *__restrict__ psd, double inv_n2,
                                                                      int in_stride, int out_stride)
{
    using PL = fft64::Plan<LOGN>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *lds = reinterpret_cast<double *>(smem);
    const int t = threadIdx.x;
    const size_t in_frame = (size_t)blockIdx.y * in_stride + blockIdx.x;
    const size_t out_frame = (size_t)blockIdx.y * out_stride + blockIdx.x;
    const float2 *x = reinterpret_cast<const float2 *>(iq) + in_frame * PL::N;

    double xr[PL::R], xi[PL::R];
#pragma unroll
    for (int m = 0; m < PL::R; m++) {
        // slot m <- x[bitrev(m) * T + t]: 512-byte contiguous segments per wave (fft_f64.h load_input)
        const int k = (int)fft64::brev_bits((unsigned)m, PL::LOGR);
        const float2 v = x[k * PL::T + t];
        xr[m] = (double)v.x;
        xi[m] = (double)v.y;
    }
    run_passes<LOGN, 0>(xr, xi, t, tw, lds);

    float *sp = spectrum + out_frame * PL::N;
    float *pd = psd + out_frame * PL::N;
#pragma unroll
    for (int s = 0; s < PL::R; s++) {
        const int i = fft64::output_bin<LOGN>(t, s);
        const int k = (i + PL::N / 2) & (PL::N - 1);                  // dsp/fft.go:54-57
        const float p = (float)(xr[s] * xr[s] + xi[s] * xi[s]);       // dsp/fft.go:71-73 PSD[float32]
        pd[k] = p;
        sp[k] = gomath::psd_value_in_db(p, inv_n2) + 120.0f;          // dsp/fft.go:79-81 + dBmShift
    }
}

// ---------------------------------------------------------------------------------------------
// FindNoiseFloor (dsp/fft.go:215-252), split so both sequential float64 chains keep their order
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_window_means(const float *__restrict__ psd, double *__restrict__ win_mean,
                                                      NoiseGeom g, int n_frames, int stride)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_frames * g.n_windows)
        return;
    const int f = idx / g.n_windows;
    const int w = idx - f * g.n_windows;
    const size_t frame = (size_t)blockIdx.y * stride + f;
    const float *p = psd + frame * g.n + g.edge + (size_t)w * g.window;
    double sum = 0;
    int i = 0;
    for (; i + 8 <= g.window; i += 8) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; k++)
            v[k] = p[i + k];
#pragma unroll
        for (int k = 0; k < 8; k++)
            sum += (double)v[k];
    }
    for (; i < g.window; i++)
        sum += (double)p[i];
    win_mean[frame * 10 + w] = sum / (double)g.window;
}

__global__ __launch_bounds__(64) void k_noise_stats(const float *__restrict__ psd, const double *__restrict__ win_mean,
                                                    sdr_frame_rec *__restrict__ recs, NoiseGeom g, int n_frames,
                                                    int stride)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_frames)
        return;
    const size_t frame = (size_t)blockIdx.y * stride + f;
    const float *p = psd + frame * g.n;
    // window selection, in the reference's order (`mean < minValue || first`, :232)
    double minValue = (double)p[0];
    bool first = true;
    double resultMean = 0;
    int resultFrom = 0, resultTo = 0;
    for (int w = 0; w < g.n_windows; w++) {
        const double mean = win_mean[frame * 10 + w];
        if (mean < minValue || first) {
            minValue = mean;
            first = false;
            resultMean = mean;
            resultFrom = g.edge;  // `from` is only ever assigned at the first iteration
            resultTo = g.edge + (w + 1) * g.window;
        }
    }
    // variance over psd[resultFrom..resultTo] about the winning mean, divided by windowSize (:244-249)
    double sum = 0;
    int i = resultFrom;
    for (; i + 8 <= resultTo + 1; i += 8) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; k++)
            v[k] = p[i + k];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const double d = (double)v[k] - resultMean;
            sum += d * d;
        }
    }
    for (; i <= resultTo; i++) {
        const double d = (double)p[i] - resultMean;
        sum += d * d;
    }
    const double variance = sum / (double)g.window;
    const float psdNoiseFloor = (float)minValue;
    sdr_frame_rec r;
    r.min_mean = psdNoiseFloor;
    r.variance = variance;
    // rx/receiver.go:383  T(float64(PSDValueIndB(T(Sqrt(var)), N) + dBmShift) * 0.25)
    r.dev_in = (float)((double)(gomath::psd_value_in_db((float)::sqrt(variance), g.inv_n2) + 120.0f) * 0.25);
    // rx/receiver.go:384  PSDValueIndB(psdNoiseFloor, N) + dBmShift
    r.nf_in = gomath::psd_value_in_db(psdNoiseFloor, g.inv_n2) + 120.0f;
    r.noise_dev = r.noise_floor = r.peak_thr = r.listen_thr = 0;
    r.pad = 0;
    recs[frame] = r;
}

// ---------------------------------------------------------------------------------------------
// k_thresholds — RollingMean.Put x2 per frame (dsp/dsp.go:257-268, rx/receiver.go:383-385) in frame
// order.  The value leaving the 60-frame window at frame f is the input of frame f-60 (or the ring
// carried over from the previous batch), so only the two float32 running sums form a serial chain:
// lanes 0 and 1 of wave 0 run them side by side, everything else is data-parallel.
// ---------------------------------------------------------------------------------------------
constexpr int THR_CHUNK = 1024;

__global__ __launch_bounds__(256) void k_thresholds(sdr_frame_rec *__restrict__ recs, BandState *__restrict__ st,
                                                    int n_frames, int stride)
{
    __shared__ float s_in[2][THR_CHUNK];
    __shared__ float s_old[2][THR_CHUNK];
    __shared__ float s_sum[2][THR_CHUNK];
    __shared__ float s_ring[2][SDR_NOISE_WINDOW];
    __shared__ float s_carry[2];
    const int band = blockIdx.x;
    const int tid = threadIdx.x;
    BandState *s = &st[band];
    sdr_frame_rec *r = recs + (size_t)band * stride;
    const int next0 = s->next;
    const float peak_threshold = s->peak_threshold;
    if (tid < SDR_NOISE_WINDOW) {
        s_ring[0][tid] = s->nf_ring[tid];
        s_ring[1][tid] = s->dev_ring[tid];
    }
    if (tid == 0) {
        s_carry[0] = s->nf_sum;
        s_carry[1] = s->dev_sum;
    }
    __syncthreads();

    for (int base = 0; base < n_frames; base += THR_CHUNK) {
        const int cnt = min(THR_CHUNK, n_frames - base);
        for (int j = tid; j < cnt; j += blockDim.x) {
            const int f = base + j;
            s_in[0][j] = r[f].nf_in;
            s_in[1][j] = r[f].dev_in;
            if (f >= SDR_NOISE_WINDOW) {
                s_old[0][j] = r[f - SDR_NOISE_WINDOW].nf_in;
                s_old[1][j] = r[f - SDR_NOISE_WINDOW].dev_in;
            } else {
                const int slot = (next0 + f) % SDR_NOISE_WINDOW;
                s_old[0][j] = s_ring[0][slot];
                s_old[1][j] = s_ring[1][slot];
            }
        }
        __syncthreads();
        if (tid < 2) {
            float sum = s_carry[tid];
            const float *in = s_in[tid], *old = s_old[tid];
            float *out = s_sum[tid];
            for (int j = 0; j < cnt; j++) {
                sum = sum - old[j];  // v.sumForMean -= v.values[v.next]
                sum = sum + in[j];   // v.sumForMean += v.values[v.next]
                out[j] = sum;
            }
            s_carry[tid] = sum;
        }
        __syncthreads();
        for (int j = tid; j < cnt; j += blockDim.x) {
            const int f = base + j;
            const float noiseFloor = __fdiv_rn(s_sum[0][j], (float)SDR_NOISE_WINDOW);
            const float noiseDeviation = __fdiv_rn(s_sum[1][j], (float)SDR_NOISE_WINDOW);
            r[f].noise_floor = noiseFloor;
            r[f].noise_dev = noiseDeviation;
            r[f].peak_thr = peak_threshold + noiseFloor;  // rx/receiver.go:385
            r[f].listen_thr = noiseFloor + noiseDeviation;  // rx/receiver.go:394
        }
        __syncthreads();
    }
    // carry the window over to the next batch: slot (next0+f)%60 holds the last input written there
    if (tid < SDR_NOISE_WINDOW) {
        // last frame f in [0,n_frames) with (next0 + f) % 60 == tid
        const int off = (tid - next0 % SDR_NOISE_WINDOW + SDR_NOISE_WINDOW) % SDR_NOISE_WINDOW;  // smallest f
        if (off < n_frames) {
            const int f = off + ((n_frames - 1 - off) / SDR_NOISE_WINDOW) * SDR_NOISE_WINDOW;
            s->nf_ring[tid] = r[f].nf_in;
            s->dev_ring[tid] = r[f].dev_in;
        }
    }
    if (tid == 0) {
        s->nf_sum = s_carry[0];
        s->dev_sum = s_carry[1];
        s->next = (next0 + n_frames) % SDR_NOISE_WINDOW;
    }
}

// ---------------------------------------------------------------------------------------------
// k_listen — one wave per tracked signal.  rx/receiver.go:388-402 -> rx/listener.go:142-148 ->
// cw/spectral.go:48-54 -> dsp/dsp.go:164-182 -> cw/decode.go:202-250.
// The 64 lanes gather spectrum[f][bin] for 64 consecutive frames and compare against that frame's
// threshold (one ballot = 64 raw on/off states); the debouncer + Morse timing state machine is
// inherently serial per signal and runs wave-uniformly over those 64 bits.
// ---------------------------------------------------------------------------------------------
struct TextSink {
    uint32_t *buf;
    uint32_t count, cap, dropped;
    bool writer;
    __device__ void put(uint32_t r)
    {
        if (count < cap) {
            if (writer)
                buf[count] = r;
            count++;
        } else {
            dropped++;
        }
    }
};

__global__ __launch_bounds__(64) void k_listen(const float *__restrict__ spectrum,
                                               const sdr_frame_rec *__restrict__ recs, ListenerSlot *__restrict__ slots,
                                               const uint16_t *__restrict__ morse, uint32_t *__restrict__ text,
                                               sdr_edge *__restrict__ edges, uint64_t *__restrict__ bits,
                                               float *__restrict__ tr_values, uint8_t *__restrict__ tr_raw,
                                               uint8_t *__restrict__ tr_deb, ListenGeom g, int n_frames)
{
    const int l = blockIdx.x, band = blockIdx.y, lane = threadIdx.x;
    ListenerSlot *slot = &slots[(size_t)band * g.max_listeners + l];
    if (!slot->active)
        return;
    const int bin = slot->bin;
    cw::Debouncer deb = slot->deb;
    cw::DecoderState dec = slot->dec;
    const size_t lidx = (size_t)band * g.max_listeners + l;
    TextSink sink{text + lidx * g.text_cap, slot->text_count, (uint32_t)g.text_cap, slot->text_dropped, lane == 0};
    sdr_edge *my_edges = edges + lidx * g.edge_cap;
    uint64_t *my_bits = bits + lidx * g.bit_words;
    uint32_t n_edges = 0;
    int last_deb = slot->last_debounced;
    const float *sp = spectrum + (size_t)band * g.stride * g.n + bin;
    const sdr_frame_rec *r = recs + (size_t)band * g.stride;

    for (int f0 = 0; f0 < n_frames; f0 += 64) {
        const int f = f0 + lane;
        float v = 0.f, thr = 0.f;
        bool raw = false;
        if (f < n_frames) {
            v = sp[(size_t)f * g.n];
            thr = r[f].listen_thr;
            raw = v > thr;  // cw/spectral.go:49
        }
        const uint64_t mask = __ballot(raw);
        const int cnt = min(64, n_frames - f0);
        uint64_t dmask = 0;
        for (int j = 0; j < cnt; j++) {
            const bool st = (mask >> j) & 1ull;
            const bool d = cw::debounce(deb, st);
            if ((int)d != last_deb) {
                if (n_edges < (uint32_t)g.edge_cap && lane == 0)
                    my_edges[n_edges] = sdr_edge{(uint32_t)(g.frame_base + f0 + j), d ? 1u : 0u};
                n_edges++;
                last_deb = d;
            }
            cw::decoder_tick(dec, d, morse, sink);
            dmask |= (uint64_t)d << j;
        }
        if (lane == 0)
            my_bits[f0 >> 6] = dmask;
        if (g.trace && f < n_frames) {
            const size_t ti = ((size_t)band * g.stride + f) * g.max_listeners + l;
            tr_values[ti] = v;
            tr_raw[ti] = raw;
            tr_deb[ti] = (dmask >> lane) & 1ull;
        }
    }
    if (lane == 0) {
        slot->deb = deb;
        slot->dec = dec;
        slot->text_count = sink.count;
        slot->text_dropped = sink.dropped;
        slot->edge_count = n_edges;
        slot->last_debounced = last_deb;
    }
}

// cw.Decoder.stop for one listener (cw/decode.go:352-354)
__global__ void k_listener_stop(ListenerSlot *slot, const uint16_t *morse, uint32_t *text, int text_cap)
{
    if (threadIdx.x != 0 || !slot->active)
        return;
    cw::DecoderState dec = slot->dec;
    TextSink sink{text, slot->text_count, (uint32_t)text_cap, slot->text_dropped, true};
    cw::decoder_stop(dec, morse, sink);
    slot->dec = dec;
    slot->text_count = sink.count;
    slot->text_dropped = sink.dropped;
}

// Receiver.SetSignalDebounce on the band's current listeners (rx/receiver.go:238-244)
__global__ void k_set_debounce(ListenerSlot *slots, int n, int threshold)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && slots[i].active)
        slots[i].deb.threshold = threshold;
}

// ---------------------------------------------------------------------------------------------
// k_cumulate — cumulation[i] += spectrum[i] (rx/receiver.go:404-407): a float32 sum in frame order.
// Slot 0 continues the cumulation carried over from the previous batch; a slot that reaches 100
// frames is written out for the peak scan, an incomplete last slot becomes the next carry.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cumulate(const float *__restrict__ spectrum, const float *__restrict__ carry_in,
                                                  float *__restrict__ carry_out, float *__restrict__ cum_out,
                                                  CumGeom g)
{
    const int i4 = blockIdx.x * blockDim.x + threadIdx.x;  // group of 4 bins
    if (i4 * 4 >= g.n)
        return;
    const int slot = blockIdx.y, band = blockIdx.z;
    // frames of this slot: slot 0 takes (100 - count0) frames, later slots 100 each
    const int first_len = SDR_CUMULATION_SIZE - g.count0;
    const int begin = slot == 0 ? 0 : first_len + (slot - 1) * SDR_CUMULATION_SIZE;
    const int len = slot == 0 ? first_len : SDR_CUMULATION_SIZE;
    const int end = min(begin + len, g.n_frames);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (slot == 0 && g.count0 > 0)
        acc = reinterpret_cast<const float4 *>(carry_in + (size_t)band * g.n)[i4];
    const float4 *sp = reinterpret_cast<const float4 *>(spectrum + (size_t)band * g.stride * g.n) + i4;
    const size_t fstride = g.n / 4;
    for (int f = begin; f < end; f++) {
        const float4 v = sp[(size_t)f * fstride];
        acc.x += v.x;
        acc.y += v.y;
        acc.z += v.z;
        acc.w += v.w;
    }
    const bool complete = (begin + len) <= g.n_frames;
    if (complete) {
        // completed chunk index == slot (slot 0 completes first if it completes at all)
        reinterpret_cast<float4 *>(cum_out + ((size_t)band * g.max_chunks + slot) * g.n)[i4] = acc;
    } else {
        reinterpret_cast<float4 *>(carry_out + (size_t)band * g.n)[i4] = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// k_find_peaks — dsp.FindPeaks (dsp/fft.go:254-285) on one completed cumulation: flag bins whose
// value/100 exceeds the threshold of the completing frame, number the runs with a workgroup prefix
// sum, and let the thread that owns a run start walk it (first maximum wins, strict `<`, :270).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_find_peaks(const float *__restrict__ cum, const sdr_frame_rec *__restrict__ recs,
                                                    DevPeak *__restrict__ peaks, int *__restrict__ counts, PeakGeom g)
{
    __shared__ int s_scan[256];
    const int chunk = blockIdx.x, band = blockIdx.y, tid = threadIdx.x;
    const int first_len = SDR_CUMULATION_SIZE - g.count0;
    const int end_frame = first_len + chunk * SDR_CUMULATION_SIZE - 1;  // frame that completes this chunk
    const float thr = recs[(size_t)band * g.stride + end_frame].peak_thr;
    const float *c = cum + ((size_t)band * g.max_chunks + chunk) * g.n;
    const float size = (float)SDR_CUMULATION_SIZE;
    const int per = g.n / 256;
    const int base = tid * per;
    bool prev = base > 0 ? (__fdiv_rn(c[base - 1], size) > thr) : false;
    int starts = 0;
    for (int i = base; i < base + per; i++) {
        const bool fl = __fdiv_rn(c[i], size) > thr;
        if (fl && !prev)
            starts++;
        prev = fl;
    }
    s_scan[tid] = starts;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {  // Hillis-Steele inclusive scan
        int v = 0;
        if (tid >= off)
            v = s_scan[tid - off];
        __syncthreads();
        s_scan[tid] += v;
        __syncthreads();
    }
    int idx = s_scan[tid] - starts;
    if (tid == 255)
        counts[(size_t)band * g.max_chunks + chunk] = s_scan[255];
    prev = base > 0 ? (__fdiv_rn(c[base - 1], size) > thr) : false;
    for (int i = base; i < base + per; i++) {
        const float value = __fdiv_rn(c[i], size);
        const bool fl = value > thr;
        if (fl && !prev) {
            float best = value;
            int best_bin = i;
            int j = i + 1;
            for (; j < g.n; j++) {
                const float vj = __fdiv_rn(c[j], size);
                if (!(vj > thr))
                    break;
                if (best < vj) {
                    best = vj;
                    best_bin = j;
                }
            }
            if (idx < g.max_peaks) {
                DevPeak p;
                p.from = i;
                p.to = j - 1;  // also N-1 for a run still open at the last bin (:276-282)
                p.signal_bin = best_bin;
                p.signal_value = best;
                p.y1 = best_bin > 0 ? c[best_bin - 1] : 0.f;
                p.y2 = c[best_bin];
                p.y3 = best_bin < g.n - 1 ? c[best_bin + 1] : 0.f;
                peaks[((size_t)band * g.max_chunks + chunk) * g.max_peaks + idx] = p;
            }
            idx++;
        }
        prev = fl;
    }
}

// ---------------------------------------------------------------------------------------------
// host-callable launchers
// ---------------------------------------------------------------------------------------------
template <int LOGN>
static hipError_t launch_fft_t(const float *iq, const fft64::cplx *tw, float *spectrum, float *psd, int n_frames,
                               int n_bands, int in_stride, int out_stride, hipStream_t stream)
{
    using PL = fft64::Plan<LOGN>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fft_project<LOGN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, PL::LDS_BYTES);
        if (e != hipSuccess)
            return e;
        attr_set = true;
    }
    const double inv_n2 = 1.0 / ((double)PL::N * (double)PL::N);
    hipLaunchKernelGGL(k_fft_project<LOGN>, dim3(n_frames, n_bands), dim3(PL::T), PL::LDS_BYTES, stream, iq, tw,
                       spectrum, psd, inv_n2, in_stride, out_stride);
    return hipGetLastError();
}

hipError_t launch_fft(int logn, const float *iq, const fft64::cplx *tw, float *spectrum, float *psd, int n_frames,
                      int n_bands, int in_stride, int out_stride, hipStream_t stream)
{
    switch (logn) {
    case 9: return launch_fft_t<9>(iq, tw, spectrum, psd, n_frames, n_bands, in_stride, out_stride, stream);
    case 10: return launch_fft_t<10>(iq, tw, spectrum, psd, n_frames, n_bands, in_stride, out_stride, stream);
    case 11: return launch_fft_t<11>(iq, tw, spectrum, psd, n_frames, n_bands, in_stride, out_stride, stream);
    case 12: return launch_fft_t<12>(iq, tw, spectrum, psd, n_frames, n_bands, in_stride, out_stride, stream);
    case 13: return launch_fft_t<13>(iq, tw, spectrum, psd, n_frames, n_bands, in_stride, out_stride, stream);
    case 14: return launch_fft_t<14>(iq, tw, spectrum, psd, n_frames, n_bands, in_stride, out_stride, stream);
    default: return hipErrorInvalidValue;
    }
}

int twiddle_count(int logn)
{
    switch (logn) {
    case 9: return fft64::Plan<9>::TW_TOTAL;
    case 10: return fft64::Plan<10>::TW_TOTAL;
    case 11: return fft64::Plan<11>::TW_TOTAL;
    case 12: return fft64::Plan<12>::TW_TOTAL;
    case 13: return fft64::Plan<13>::TW_TOTAL;
    case 14: return fft64::Plan<14>::TW_TOTAL;
    default: return 0;
    }
}

void build_twiddles(int logn, const double *wre, const double *wim, fft64::cplx *out)
{
    switch (logn) {
    case 9: fft64::build_pass_twiddles<9>(wre, wim, out); break;
    case 10: fft64::build_pass_twiddles<10>(wre, wim, out); break;
    case 11: fft64::build_pass_twiddles<11>(wre, wim, out); break;
    case 12: fft64::build_pass_twiddles<12>(wre, wim, out); break;
    case 13: fft64::build_pass_twiddles<13>(wre, wim, out); break;
    case 14: fft64::build_pass_twiddles<14>(wre, wim, out); break;
    default: break;
    }
}

hipError_t launch_window_means(const float *psd, double *win_mean, NoiseGeom g, int n_frames, int n_bands, int stride,
                               hipStream_t stream)
{
    const int total = n_frames * g.n_windows;
    hipLaunchKernelGGL(k_window_means, dim3((total + 255) / 256, n_bands), dim3(256), 0, stream, psd, win_mean, g,
                       n_frames, stride);
    return hipGetLastError();
}

hipError_t launch_noise_stats(const float *psd, const double *win_mean, sdr_frame_rec *recs, NoiseGeom g, int n_frames,
                              int n_bands, int stride, hipStream_t stream)
{
    hipLaunchKernelGGL(k_noise_stats, dim3((n_frames + 63) / 64, n_bands), dim3(64), 0, stream, psd, win_mean, recs, g,
                       n_frames, stride);
    return hipGetLastError();
}

hipError_t launch_thresholds(sdr_frame_rec *recs, BandState *st, int n_frames, int n_bands, int stride,
                             hipStream_t stream)
{
    hipLaunchKernelGGL(k_thresholds, dim3(n_bands), dim3(256), 0, stream, recs, st, n_frames, stride);
    return hipGetLastError();
}

hipError_t launch_listen(const float *spectrum, const sdr_frame_rec *recs, ListenerSlot *slots, const uint16_t *morse,
                         uint32_t *text, sdr_edge *edges, uint64_t *bits, float *tr_values, uint8_t *tr_raw,
                         uint8_t *tr_deb, ListenGeom g, int n_frames, int n_slots, int n_bands, hipStream_t stream)
{
    if (n_slots == 0)
        return hipSuccess;
    hipLaunchKernelGGL(k_listen, dim3(n_slots, n_bands), dim3(64), 0, stream, spectrum, recs, slots, morse, text, edges,
                       bits, tr_values, tr_raw, tr_deb, g, n_frames);
    return hipGetLastError();
}

hipError_t launch_listener_stop(ListenerSlot *slot, const uint16_t *morse, uint32_t *text, int text_cap,
                                hipStream_t stream)
{
    hipLaunchKernelGGL(k_listener_stop, dim3(1), dim3(64), 0, stream, slot, morse, text, text_cap);
    return hipGetLastError();
}

hipError_t launch_set_debounce(ListenerSlot *slots, int n, int threshold, hipStream_t stream)
{
    hipLaunchKernelGGL(k_set_debounce, dim3((n + 63) / 64), dim3(64), 0, stream, slots, n, threshold);
    return hipGetLastError();
}

hipError_t launch_cumulate(const float *spectrum, const float *carry_in, float *carry_out, float *cum_out, CumGeom g,
                           int n_slots, int n_bands, hipStream_t stream)
{
    hipLaunchKernelGGL(k_cumulate, dim3((g.n / 4 + 255) / 256, n_slots, n_bands), dim3(256), 0, stream, spectrum,
                       carry_in, carry_out, cum_out, g);
    return hipGetLastError();
}

hipError_t launch_find_peaks(const float *cum, const sdr_frame_rec *recs, DevPeak *peaks, int *counts, PeakGeom g,
                             int n_chunks, int n_bands, hipStream_t stream)
{
    if (n_chunks == 0)
        return hipSuccess;
    hipLaunchKernelGGL(k_find_peaks, dim3(n_chunks, n_bands), dim3(256), 0, stream, cum, recs, peaks, counts, g);
    return hipGetLastError();
}

}  // namespace sdr
