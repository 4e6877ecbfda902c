// sdr_audio.hip — the audio path of the reference (cw.AudioDemodulator, cw/audio.go:21-211) for a
// batch of independent mono streams: per-block Goertzel magnitude ("mix to the pitch, boxcar-decimate
// by blocksize, envelope magnitude", dsp/dsp.go:34-136) is data-parallel over blocks; the stateful
// magnitude normalisation, debouncer and Morse decoder run once per stream, in block order.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/sdrainer_hip.h"
#include "cw_decoder.h"
#include "gomath.h"

namespace sdr {
int set_error(int code, const char *msg);  // capi_bank.hip: feeds sdr_last_error()
}

namespace {

struct GoertzelParams {
    int blocksize;
    double coeff;
    double magnitudeLimitLow;
    double maxScale;  // cw/audio.go:19 defaultMaxScale
    float scale;      // 0 = autoscale (cw/audio.go:183-187)
};

struct AudioStreamState {
    double magnitudeLimit;      // dsp/dsp.go:41
    double magnitudeThreshold;  // dsp/dsp.go:42
    cw::Debouncer deb;
    cw::DecoderState dec;
    uint32_t text_count, text_dropped;
};

// cw/audio.go:213-221 truncate
__device__ __forceinline__ float truncate1(float v) { return v > 1.f ? 1.f : (v < -1.f ? -1.f : v); }

// One lane per (stream, block): cw/audio.go:181-193 scaling + dsp/dsp.go:98-106 Goertzel.Magnitude
__global__ __launch_bounds__(64) void k_goertzel_blocks(const float *__restrict__ samples, double *__restrict__ mags,
                                                        GoertzelParams g, int n_blocks, int sample_stride,
                                                        int block_stride)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks)
        return;
    const int stream = blockIdx.y;
    const float *x = samples + (size_t)stream * sample_stride + (size_t)b * g.blocksize;
    float scale = g.scale;
    if (scale == 0.f) {
        float mx = 0.f;  // dsp/dsp.go:19-28 FilterBlock.Max
        for (int i = 0; i < g.blocksize; i++) {
            const float a = (float)::fabs((double)x[i]);
            if (a > mx)
                mx = a;
        }
        const double inv = 1 / (double)mx;
        scale = (float)(inv < g.maxScale ? inv : g.maxScale);  // math.Min(1/max, maxScale)
    }
    double q1 = 0, q2 = 0;
    for (int i = 0; i < g.blocksize; i++) {
        float s = x[i];
        if (scale != 1.f)
            s = truncate1(s * scale);
        const double q0 = g.coeff * q1 - q2 + (double)s;
        q2 = q1;
        q1 = q0;
    }
    mags[(size_t)stream * block_stride + b] = ::sqrt((q1 * q1) + (q2 * q2) - q1 * q2 * g.coeff);
}

struct AudioSink {
    uint32_t *buf;
    uint32_t count, cap, dropped;
    __device__ void put(uint32_t r)
    {
        if (count < cap)
            buf[count++] = r;
        else
            dropped++;
    }
};

// One thread per stream, in block order: dsp/dsp.go:111-136 NormalizedMagnitude + Detect,
// cw/audio.go:202-203 debounce + Decoder.Tick.
__global__ void k_audio_decode(double *__restrict__ mags, AudioStreamState *__restrict__ st,
                               const uint16_t *__restrict__ morse, uint32_t *__restrict__ text,
                               uint8_t *__restrict__ raw, uint8_t *__restrict__ deb, GoertzelParams g, int n_blocks,
                               int block_stride, int text_cap, int n_streams, int flush)
{
    const int stream = blockIdx.x * blockDim.x + threadIdx.x;
    if (stream >= n_streams)
        return;
    AudioStreamState s = st[stream];
    AudioSink sink{text + (size_t)stream * text_cap, s.text_count, (uint32_t)text_cap, s.text_dropped};
    double *m = mags + (size_t)stream * block_stride;
    for (int b = 0; b < n_blocks; b++) {
        const double magnitude = m[b];
        if (magnitude > g.magnitudeLimitLow)
            s.magnitudeLimit = (s.magnitudeLimit + ((magnitude - s.magnitudeLimit) / 6));
        if (s.magnitudeLimit < g.magnitudeLimitLow)
            s.magnitudeLimit = g.magnitudeLimitLow;
        const double norm = magnitude / s.magnitudeLimit;
        const bool state = norm > s.magnitudeThreshold;
        const bool d = cw::debounce(s.deb, state);
        cw::decoder_tick(s.dec, d, morse, sink);
        m[b] = norm;
        raw[(size_t)stream * block_stride + b] = state;
        deb[(size_t)stream * block_stride + b] = d;
    }
    if (flush)
        cw::decoder_stop(s.dec, morse, sink);  // cw/audio.go:205-207
    s.text_count = sink.count;
    s.text_dropped = sink.dropped;
    st[stream] = s;
}

int afail(int code, const std::string &msg) { return sdr::set_error(code, msg.c_str()); }
#define AHIP(expr)                                                                          \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess)                                                               \
            return afail(SDR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));   \
    } while (0)

}  // namespace

struct sdr_audio {
    int n_streams = 0, sample_rate = 0, max_blocks = 0, device = 0, text_cap = 4096;
    GoertzelParams g{};
    std::vector<std::vector<float>> pending;  // per-stream samples not yet forming a whole block
    float *d_samples = nullptr;
    double *d_mags = nullptr;
    uint8_t *d_raw = nullptr, *d_deb = nullptr;
    AudioStreamState *d_state = nullptr;
    uint16_t *d_morse = nullptr;
    uint32_t *d_text = nullptr;
    int last_blocks = 0;
    std::vector<float> h_samples;
};

extern "C" {
#pragma GCC visibility push(default)

int sdr_audio_create(int n_streams, double pitch, int sample_rate, int max_blocks, int device_id, sdr_audio **out)
{
    if (!out || n_streams < 1 || sample_rate < 1 || max_blocks < 1 || !(pitch > 0))
        return afail(SDR_ERR_BAD_ARG, "bad audio geometry");
    AHIP(hipSetDevice(device_id));
    sdr_audio *a = new sdr_audio();
    a->n_streams = n_streams;
    a->sample_rate = sample_rate;
    a->max_blocks = max_blocks;
    a->device = device_id;
    // dsp.NewGoertzel (dsp/dsp.go:55-70) with DefaultBlocksizeRatio (dsp/dsp.go:11)
    const double ratio = 0.005;
    const double minBlocksize = std::round((double)sample_rate / pitch);  // calculateBlocksize :72-75
    a->g.blocksize = (int)std::round((ratio * (double)sample_rate) / minBlocksize) * (int)minBlocksize;
    if (a->g.blocksize < 1) {
        delete a;
        return afail(SDR_ERR_BAD_ARG, "pitch / sample rate give an empty Goertzel block");
    }
    const int binIndex = (int)(0.5 + ((double)a->g.blocksize * pitch / (double)sample_rate));
    const double omega = 2 * 3.14159265358979323846264338327950288 * (double)binIndex / (double)a->g.blocksize;
    double sn, cs;
    gomath::sincos(omega, &sn, &cs);  // math.Cos shares Sincos' reduction and polynomial
    a->g.coeff = 2 * cs;
    a->g.magnitudeLimitLow = (double)a->g.blocksize / 2;
    a->g.maxScale = 12;
    a->g.scale = 1.f;  // NewAudioDemodulator, cw/audio.go:45
    a->pending.resize((size_t)n_streams);

    // (a failure half way frees what was allocated so far: sdr_audio_destroy tolerates null members)
#define ACREATE(expr)                                                                       \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            const std::string _msg = std::string(#expr) + ": " + hipGetErrorString(_e);     \
            sdr_audio_destroy(a);                                                           \
            return afail(SDR_ERR_HIP, _msg);                                                \
        }                                                                                   \
    } while (0)
    const size_t S = (size_t)n_streams, Bk = (size_t)max_blocks;
    ACREATE(hipMalloc((void **)&a->d_samples, sizeof(float) * S * Bk * (size_t)a->g.blocksize));
    ACREATE(hipMalloc((void **)&a->d_mags, sizeof(double) * S * Bk));
    ACREATE(hipMalloc((void **)&a->d_raw, S * Bk));
    ACREATE(hipMalloc((void **)&a->d_deb, S * Bk));
    ACREATE(hipMalloc((void **)&a->d_state, sizeof(AudioStreamState) * S));
    ACREATE(hipMalloc((void **)&a->d_morse, sizeof(uint16_t) * cw::kMorseTableSize));
    ACREATE(hipMalloc((void **)&a->d_text, sizeof(uint32_t) * S * (size_t)a->text_cap));
    std::vector<uint16_t> morse(cw::kMorseTableSize);
    cw::build_morse_table(morse.data());
    ACREATE(hipMemcpy(a->d_morse, morse.data(), sizeof(uint16_t) * cw::kMorseTableSize, hipMemcpyHostToDevice));
    std::vector<AudioStreamState> st(S);
    for (auto &s : st) {
        memset(&s, 0, sizeof s);
        s.magnitudeLimit = 0;
        s.magnitudeThreshold = 0.75;                                    // dsp/dsp.go:12
        cw::debouncer_init(s.deb, 3);                                   // cw/audio.go:18
        cw::decoder_init(s.dec, sample_rate, a->g.blocksize);           // cw/audio.go:53
    }
    ACREATE(hipMemcpy(a->d_state, st.data(), sizeof(AudioStreamState) * S, hipMemcpyHostToDevice));
#undef ACREATE
    *out = a;
    return SDR_OK;
}

int sdr_audio_destroy(sdr_audio *a)
{
    if (!a)
        return SDR_OK;
    (void)hipSetDevice(a->device);
    (void)hipDeviceSynchronize();
    (void)hipFree(a->d_samples);
    (void)hipFree(a->d_mags);
    (void)hipFree(a->d_raw);
    (void)hipFree(a->d_deb);
    (void)hipFree(a->d_state);
    (void)hipFree(a->d_morse);
    (void)hipFree(a->d_text);
    delete a;
    return SDR_OK;
}

int sdr_audio_blocksize(sdr_audio *a) { return a ? a->g.blocksize : -1; }

int sdr_audio_set_scale(sdr_audio *a, double scale)
{
    if (!a)
        return afail(SDR_ERR_BAD_ARG, "null");
    a->g.scale = (float)scale;
    return SDR_OK;
}

static int update_states(sdr_audio *a, int debounce, double threshold, bool set_deb, bool set_thr)
{
    std::vector<AudioStreamState> st((size_t)a->n_streams);
    AHIP(hipSetDevice(a->device));
    AHIP(hipMemcpy(st.data(), a->d_state, sizeof(AudioStreamState) * st.size(), hipMemcpyDeviceToHost));
    for (auto &s : st) {
        if (set_deb)
            s.deb.threshold = debounce;
        if (set_thr)
            s.magnitudeThreshold = threshold;
    }
    AHIP(hipMemcpy(a->d_state, st.data(), sizeof(AudioStreamState) * st.size(), hipMemcpyHostToDevice));
    return SDR_OK;
}

int sdr_audio_set_debounce(sdr_audio *a, int threshold)
{
    if (!a)
        return afail(SDR_ERR_BAD_ARG, "null");
    return update_states(a, threshold, 0, true, false);
}

int sdr_audio_set_magnitude_threshold(sdr_audio *a, double t)
{
    if (!a)
        return afail(SDR_ERR_BAD_ARG, "null");
    return update_states(a, 0, t, false, true);
}

static int run_blocks(sdr_audio *a, int n_blocks, int flush)
{
    AHIP(hipSetDevice(a->device));
    const int bs = a->g.blocksize;
    if (n_blocks > 0) {
        AHIP(hipMemcpy(a->d_samples, a->h_samples.data(), sizeof(float) * a->h_samples.size(), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_goertzel_blocks, dim3((n_blocks + 63) / 64, a->n_streams), dim3(64), 0, 0, a->d_samples,
                           a->d_mags, a->g, n_blocks, n_blocks * bs, a->max_blocks);
        AHIP(hipGetLastError());
    }
    hipLaunchKernelGGL(k_audio_decode, dim3((a->n_streams + 63) / 64), dim3(64), 0, 0, a->d_mags, a->d_state,
                       a->d_morse, a->d_text, a->d_raw, a->d_deb, a->g, n_blocks, a->max_blocks, a->text_cap,
                       a->n_streams, flush);
    AHIP(hipGetLastError());
    AHIP(hipDeviceSynchronize());
    a->last_blocks = n_blocks;
    return SDR_OK;
}

int sdr_audio_write(sdr_audio *a, const float *samples, int n_samples)
{
    if (!a || !samples || n_samples < 0)
        return afail(SDR_ERR_BAD_ARG, "bad audio write");
    const int bs = a->g.blocksize;
    // every stream receives the same number of samples, so they complete blocks in lockstep
    const int have = (int)a->pending[0].size() + n_samples;
    const int n_blocks = have / bs;
    if (n_blocks > a->max_blocks)
        return afail(SDR_ERR_WOULD_DROP, "more blocks than max_blocks in one write");
    a->h_samples.assign((size_t)a->n_streams * (size_t)n_blocks * (size_t)bs, 0.f);
    for (int s = 0; s < a->n_streams; s++) {
        std::vector<float> &p = a->pending[(size_t)s];
        p.insert(p.end(), samples + (size_t)s * n_samples, samples + (size_t)(s + 1) * n_samples);
        const size_t used = (size_t)n_blocks * (size_t)bs;
        if (used)
            memcpy(a->h_samples.data() + (size_t)s * used, p.data(), sizeof(float) * used);
        p.erase(p.begin(), p.begin() + (long)used);
    }
    if (n_blocks == 0) {
        a->last_blocks = 0;
        return SDR_OK;
    }
    return run_blocks(a, n_blocks, 0);
}

int sdr_audio_close(sdr_audio *a)
{
    if (!a)
        return afail(SDR_ERR_BAD_ARG, "null");
    a->h_samples.clear();
    return run_blocks(a, 0, 1);
}

int sdr_audio_read_text(sdr_audio *a, int stream, char *out, int max_bytes, int *n_bytes)
{
    if (!a || stream < 0 || stream >= a->n_streams)
        return afail(SDR_ERR_BAD_ARG, "stream out of range");
    AHIP(hipSetDevice(a->device));
    AudioStreamState s;
    AHIP(hipMemcpy(&s, a->d_state + stream, sizeof s, hipMemcpyDeviceToHost));
    std::vector<uint32_t> runes(s.text_count);
    if (s.text_count)
        AHIP(hipMemcpy(runes.data(), a->d_text + (size_t)stream * a->text_cap, sizeof(uint32_t) * s.text_count,
                       hipMemcpyDeviceToHost));
    int n = 0;
    for (uint32_t r : runes) {
        char tmp[4];
        size_t k;
        if (r < 0x80) {
            tmp[0] = (char)r;
            k = 1;
        } else if (r < 0x800) {
            tmp[0] = (char)(0xC0 | (r >> 6));
            tmp[1] = (char)(0x80 | (r & 0x3F));
            k = 2;
        } else {
            tmp[0] = (char)(0xE0 | (r >> 12));
            tmp[1] = (char)(0x80 | ((r >> 6) & 0x3F));
            tmp[2] = (char)(0x80 | (r & 0x3F));
            k = 3;
        }
        if (n + (int)k > max_bytes)
            break;
        if (out)
            memcpy(out + n, tmp, k);
        n += (int)k;
    }
    if (n_bytes)
        *n_bytes = n;
    const uint32_t zero = 0;
    AHIP(hipMemcpy(&a->d_state[stream].text_count, &zero, sizeof zero, hipMemcpyHostToDevice));
    return SDR_OK;
}

int sdr_audio_read_trace(sdr_audio *a, int stream, double *magnitudes, uint8_t *raw, uint8_t *debounced, int max,
                         int *n_blocks)
{
    if (!a || stream < 0 || stream >= a->n_streams)
        return afail(SDR_ERR_BAD_ARG, "stream out of range");
    AHIP(hipSetDevice(a->device));
    const int n = a->last_blocks < max ? a->last_blocks : max;
    if (n_blocks)
        *n_blocks = a->last_blocks;
    if (n <= 0)
        return SDR_OK;
    const size_t off = (size_t)stream * a->max_blocks;
    if (magnitudes)
        AHIP(hipMemcpy(magnitudes, a->d_mags + off, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
    if (raw)
        AHIP(hipMemcpy(raw, a->d_raw + off, (size_t)n, hipMemcpyDeviceToHost));
    if (debounced)
        AHIP(hipMemcpy(debounced, a->d_deb + off, (size_t)n, hipMemcpyDeviceToHost));
    return SDR_OK;
}

#pragma GCC visibility pop
}  // extern "C"
