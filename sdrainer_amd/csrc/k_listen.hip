// k_listen.hip — the per-signal chain: rx/receiver.go:388-402 -> rx/listener.go:142-148 ->
// cw/spectral.go:48-54 (value > threshold) -> dsp/dsp.go:164-182 (debounce) -> cw/decode.go:202-250
// (Morse timing state machine).  Compiled with -ffp-contract=off.
//
// Two kernels:
//   k_listen_gather  data-parallel over (signal, frame): one lane takes 64 consecutive frames of one
//                    signal, reads the psd values the FFT kernel tapped for it, projects them to dB,
//                    compares against each frame's threshold and packs the 64 results into one word.
//   k_listen_decode  one LANE per signal (4 signals per wave): the debouncer and the decoder are
//                    inherently serial per signal, but between keying edges Decoder.Tick only counts, so
//                    the lane walks RUNS of equal bits (ffs on the XOR-ed word) and handles each run in
//                    closed form (cw::decoder_advance) — a few hundred edges per batch instead of
//                    thousands of ticks.
#include <hip/hip_runtime.h>

#include "../../include/sdrainer_hip.h"
#include "cw_decoder.h"
#include "gomath.h"
#include "sdr_device.h"

namespace sdr {

// One lane per listener slot, one wave per 64 frames: lane l reads tap[f][l] for the wave's frames (the FFT kernel
// wrote the psd value of the slot's bin there, k_fft_psd.hip "The tap"; neighbouring lanes read neighbouring
// words), projects it to dB - the certified table shortcut of gomath.h, the literal Go algorithm where its
// certificate fails - and collects its own 64 comparison results into one word.  Sixteen waves to a workgroup: round
// 2's one-wave workgroups with the literal logarithm (0.031 ms) went to 128 different CUs and kept an FFT workgroup,
// which needs a whole CU, off each of them for that long; these are 8 workgroups for a few microseconds.
// Waves per workgroup.  Four decoder waves (one per SIMD) share a workgroup: with every lane of a wave working
// (k_listen_decode) they run as fast together as alone - 0.134 against 0.130 ms - and 16 workgroups keep 16 CUs from
// the FFT instead of 64: 0.221 against 0.234 ms per pipelined step (eight waves: 0.159 ms / 0.247, sixteen: 0.240 /
// 0.32; before the lanes were filled, four waves of four active lanes took 0.234 ms against 0.137 alone).  The gather
// waves stay one to a workgroup (two, four, eight: no difference beyond the noise).
#ifndef SDR_GATHER_WAVES
#define SDR_GATHER_WAVES 16
#endif
#ifndef SDR_DECODE_WAVES
#define SDR_DECODE_WAVES 4
#endif
constexpr int GATHER_WAVES = SDR_GATHER_WAVES;  // 64-frame words per workgroup, one per wave
constexpr int DECODE_WAVES = SDR_DECODE_WAVES;  // signal groups per workgroup, one per wave

// the literal Go algorithm, out of line: it is rare (about three values in 10^5 fail the shortcut's certificate)
__device__ __attribute__((noinline)) float gather_db_slow(float psd, double inv_n2) { return gomath::psd_value_in_db(psd, inv_n2); }

__global__ __launch_bounds__(64 * GATHER_WAVES) void k_listen_gather(const float *__restrict__ tap, const float *__restrict__ psd,
                                                      const sdr_frame_rec *__restrict__ recs,
                                                      const ListenerSlot *__restrict__ slots, const void *__restrict__ db_tab,
                                                      uint64_t *__restrict__ raw_bits, float *__restrict__ tr_values,
                                                      uint8_t *__restrict__ tr_raw, const BatchCursor *__restrict__ cur, ListenGeom g,
                                                      int n_frames, int n_slots, double inv_n2)
{
    if (cur)
        g.frame_base = cur->frame_base;
    // the tables of the certified dB shortcut (gomath.h; k_cumulate uses the same ones)
    __shared__ __attribute__((aligned(16))) unsigned char s_tab[gomath::kDbTabBytes];
    {
        const uint4 *src = static_cast<const uint4 *>(db_tab);
        uint4 *dst = reinterpret_cast<uint4 *>(s_tab);
        for (int i = threadIdx.x; i < gomath::kDbTabBytes / 16; i += blockDim.x)
            dst[i] = src[i];
    }
    __syncthreads();
    const gomath::DbTables tab = gomath::db_tables(s_tab);
    const int word = blockIdx.x * GATHER_WAVES + (int)(threadIdx.x >> 6), band = blockIdx.z;
    const int l = blockIdx.y * 64 + (int)(threadIdx.x & 63);
    if (l >= n_slots || word * 64 >= n_frames)
        return;
    const size_t lidx = (size_t)band * g.max_listeners + l;
    if (!slots[lidx].active)
        return;
    const int f0 = word * 64;
    const int cnt = min(64, n_frames - f0);
    const size_t frame0 = (size_t)band * g.stride + f0;
    // a listener bound to this batch after its FFT ran (sdr_attach_at): frames of the batch before `skip` are not its
    // own, frames before `untapped` have no tap entry - their value comes from the retained psd row
    const int skip = (int)(slots[lidx].start_frame - g.frame_base), untapped = (int)(slots[lidx].tapped_from - g.frame_base);
    const int bin = slots[lidx].bin;
    uint64_t mask = 0;
    for (int j0 = 0; j0 < cnt; j0 += 8) {
        float p[8], thr[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int jj = min(j0 + k, cnt - 1);
            p[k] = f0 + jj < untapped ? psd[(frame0 + jj) * (size_t)g.n + bin] : tap[(frame0 + jj) * g.max_listeners + l];
            thr[k] = recs[frame0 + jj].listen_thr;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int j = j0 + k;
            if (j < cnt) {
                float db;
                if (!gomath::psd_value_in_db_fast(p[k], tab, &db))
                    db = gather_db_slow(p[k], inv_n2);
                const float v = db + (float)SDR_DBM_SHIFT;  // spectrum[SignalBin]
                const bool raw = v > thr[k];                // cw/spectral.go:49
                mask |= (uint64_t)raw << j;
                if (g.trace) {
                    const size_t ti = (frame0 + j) * g.max_listeners + l;
                    tr_values[ti] = v;
                    tr_raw[ti] = raw;
                }
            }
        }
    }
    if (skip > f0)  // (the decoder starts at `skip` too; the word stays clean for readers of the raw states)
        mask &= skip - f0 >= 64 ? 0ull : ~((1ull << (skip - f0)) - 1ull);
    raw_bits[lidx * g.bit_words + word] = mask;
}

// Signals per decoder wave.  The lanes of a wave walk their signals' edges in lockstep and take every branch any of
// them takes, so fewer signals per wave means a shorter wave - and the decoder is the long pole of the listen stream.
// Standalone / per pipelined step, four waves per workgroup: 1 signal 0.089 ms / 0.2327, 2: 0.115 / 0.2267,
// 4: 0.134 / 0.2207, 8: 0.151 / 0.2349 (fewer signals per wave = more workgroups holding CUs the FFT wants).
// The kernel is compiled for 1, 2 and 4 signals per wave and the launch picks the fewest that keep the decoders
// within 64 waves (16 workgroups): a small pool - config 2's 16 signals - gets a wave per signal (0.39 instead of
// 0.59 ms per 4096-frame batch), config 3's 256 signals get four per wave.
#ifndef SDR_DECODE_LANES
#define SDR_DECODE_LANES 4
#endif
constexpr int DECODE_LANES_MAX = SDR_DECODE_LANES;

// The io.Writer of a listener's decoder (cw/decode.go:352): runes and, beside each, the bank frame index of the
// Tick that wrote it (the host stamps TextProcessor.Write with that frame's time: rx/text_processor.go:208-209,
// which is what the listener's silence time-out is measured from, rx/listener.go:126-136).
struct TextSink {
    uint32_t *buf;
    uint32_t *frames;
    uint32_t count, cap, dropped;
    uint32_t frame;     // frame of the tick being processed
    uint32_t run_base;  // frame of the first tick of the run decoder_advance is walking
    bool writer;        // false: a lane that only follows another lane's signal (k_listen_decode) and stores nothing
    __device__ void at_run_tick(int k) { frame = run_base + (uint32_t)k; }
    __device__ void put(uint32_t r)
    {
        if (count < cap) {
            if (writer) {
                buf[count] = r;
                frames[count] = frame;
            }
            count++;
        } else {
            dropped++;
        }
    }
};

// (the decoder as the kernel runs it - one place per loop iteration that writes runes - is cw_decoder.h's
// decoder_run / decoder_edge_deferred, checked tick by tick against Decoder.Tick on the CPU: tests/emu/emu_decoder.cpp)
using cw::Emission;
using cw::kInvalidChar;

template <int DECODE_LANES>
__global__ __launch_bounds__(64 * DECODE_WAVES) void k_listen_decode(ListenerSlot *__restrict__ slots, const uint16_t *morse,
                                                      const uint64_t *__restrict__ raw_bits,
                                                      uint64_t *__restrict__ deb_bits, uint32_t *__restrict__ text,
                                                      uint32_t *__restrict__ text_frames, sdr_edge *__restrict__ edges,
                                                      uint32_t *__restrict__ edge_counts, uint8_t *__restrict__ tr_deb,
                                                      DropCounters *__restrict__ drops,
                                                      const BatchCursor *__restrict__ cur, ListenGeom g, int n_frames,
                                                      int n_total)
{
    if (cur)
        g.frame_base = cur->frame_base;
    // the 512-entry code table is hit on every decoded character, on the serial path: keep it in LDS
    __shared__ uint16_t s_morse[cw::kMorseTableSize];
    for (int i = threadIdx.x; i < cw::kMorseTableSize; i += blockDim.x)
        s_morse[i] = morse[i];
    __syncthreads();
    morse = s_morse;
    // Every lane of the wave works: lanes DECODE_LANES.. follow the signal of lane (lane mod DECODE_LANES) - same
    // loads, same arithmetic, same branches, no stores.  A wave with a few active lanes is the slow case of this
    // hardware, and several such waves on one CU slow each other further (tools/ubench_share.hip, dependent
    // v_add_f64: 1.68 ms with 4 active lanes against 1.31 ms with 64, one wave per CU; 3.06 against 1.31 ms with four
    // waves per CU, 4.60 against 1.31 with eight).  Full waves cost the same however many share the CU, so the
    // decoders can sit DECODE_WAVES to a workgroup and leave the other CUs to the FFT.
    const int lane = threadIdx.x & 63;
    const int first = (blockIdx.x * DECODE_WAVES + (int)(threadIdx.x >> 6)) * DECODE_LANES;  // (band, slot) flattened
    int sub = lane % DECODE_LANES;
    const bool mine = lane < DECODE_LANES && first + sub < n_total && slots[min(first + sub, n_total - 1)].active;
    const unsigned long long live = __ballot(mine);  // bit i: signal first+i is decoded by this wave
    if (!live)
        return;
    if (!((live >> sub) & 1ull))
        sub = __ffsll((long long)live) - 1;  // nothing of its own to follow: follow the wave's first signal
    const bool writer = mine;
    const int idx = first + sub;
    ListenerSlot *slot = &slots[idx];
    cw::Debouncer deb = slot->deb;
    cw::DecoderState dec = slot->dec;
    TextSink sink{text + (size_t)idx * g.text_cap, text_frames + (size_t)idx * g.text_cap, slot->text_count,
                  (uint32_t)g.text_cap, slot->text_dropped, g.frame_base, g.frame_base, writer};
    sdr_edge *my_edges = edges + (size_t)idx * g.edge_cap;
    const uint64_t *rw = raw_bits + (size_t)idx * g.bit_words;
    uint64_t *dw = deb_bits + (size_t)idx * g.bit_words;
    uint32_t n_edges = 0;
    const int band = idx / g.max_listeners, l = idx - band * g.max_listeners;
    // frames of this batch before the listener's first one (sdr_attach_at; 0 or less for everybody else)
    const int skip = (int)(slot->start_frame - g.frame_base);

    // The words of raw states are fetched four ahead: a load per word in the loop would put a trip to memory on the
    // serial path of every 64 frames (32 of them per 2048-frame batch).
    const int n_words = (n_frames + 63) >> 6;
    uint64_t ahead[4];
#pragma unroll
    for (int k = 0; k < 4; k++)
        ahead[k] = rw[min(k, n_words - 1)];
    for (int f0 = 0; f0 < n_frames; f0 += 64) {
        const int cnt = min(64, n_frames - f0);
        const uint64_t raw = ahead[0];
        ahead[0] = ahead[1];
        ahead[1] = ahead[2];
        ahead[2] = ahead[3];
        ahead[3] = rw[min((f0 >> 6) + 4, n_words - 1)];
        int pos = skip > f0 ? min(skip - f0, cnt) : 0;  // ticks of this word that are not the listener's
        // dsp/dsp.go:164-182, a run of equal raw states at a time
        const uint64_t d = pos >= cnt ? 0ull : pos ? cw::debounce_word(deb, raw >> pos, cnt - pos) << pos : cw::debounce_word(deb, raw, cnt);
        if (writer)
            dw[f0 >> 6] = d;
        // walk the runs of equal debounced bits
        while (pos < cnt) {
            const bool cur = dec.lastState != 0;
            uint64_t diff = (cur ? ~d : d) >> pos;  // 1 where the bit differs from the decoder's state
            if (cnt - pos < 64)
                diff &= (1ull << (cnt - pos)) - 1ull;
            const int run = diff ? (__ffsll((long long)diff) - 1) : (cnt - pos);
            Emission em{0u, 0u, false};
            cw::decoder_run(dec, run, g.frame_base + (uint32_t)(f0 + pos), em);
            pos += run;
            uint32_t edge_frame = 0;
            if (pos < cnt) {  // the edge tick
                const bool st = !cur;
                edge_frame = g.frame_base + (uint32_t)(f0 + pos);
                if (writer && n_edges < (uint32_t)g.edge_cap)
                    my_edges[n_edges] = sdr_edge{edge_frame, st ? 1u : 0u};
                n_edges++;
                cw::decoder_edge_deferred(dec, st, edge_frame, em);
                pos++;
            }
            if (em.key | (uint32_t)em.space) {  // the one place a tick's runes are written
                if (em.key) {
                    uint32_t r = cw::kUnknownCharacter;
                    if (em.key != kInvalidChar) {
                        const uint32_t looked_up = morse[em.key];
                        r = looked_up ? looked_up : cw::kUnknownCharacter;
                    }
                    sink.frame = em.frame;
                    sink.put(r);
                }
                if (em.space) {
                    sink.frame = edge_frame;
                    sink.put(' ');
                }
            }
        }
        if (g.trace && writer)
            for (int j = 0; j < cnt; j++)
                tr_deb[((size_t)band * g.stride + f0 + j) * g.max_listeners + l] = (d >> j) & 1ull;
    }
    if (!writer)
        return;
    slot->deb = deb;
    slot->dec = dec;
    // Frame numbers are 32 bits and compared as differences: a listener that has started must not keep a start_frame that
    // falls 2^31 frames behind (config 5 gets there in 100 days and a decode-mode listener has no time-out) - the
    // difference would turn positive and the listener go deaf.  Once a batch has reached the listener's first frame, both
    // marks move along with the batches (nothing per listener depends on the absolute number after that).
    if (skip < n_frames)
        slot->start_frame = slot->tapped_from = g.frame_base + (uint32_t)n_frames;
    slot->text_count = sink.count;
    // nothing is lost silently: what did not fit is counted bank-wide (rare; the host reads or polls the totals)
    if (sink.dropped != slot->text_dropped)
        atomicAdd(&drops->runes, (unsigned long long)(sink.dropped - slot->text_dropped));
    if (n_edges > (uint32_t)g.edge_cap)
        atomicAdd(&drops->edges, (unsigned long long)(n_edges - (uint32_t)g.edge_cap));
    slot->text_dropped = sink.dropped;
    edge_counts[idx] = n_edges;
}

// cw.Decoder.stop for one listener (cw/decode.go:352-354)
__global__ void k_listener_stop(ListenerSlot *slot, const uint16_t *morse, uint32_t *text, uint32_t *text_frames, int text_cap,
                                uint32_t frame, DropCounters *drops)
{
    if (threadIdx.x != 0 || !slot->active)
        return;
    cw::DecoderState dec = slot->dec;
    TextSink sink{text, text_frames, slot->text_count, (uint32_t)text_cap, slot->text_dropped, frame, frame, true};
    cw::decoder_stop(dec, morse, sink);
    slot->dec = dec;
    slot->text_count = sink.count;
    if (sink.dropped != slot->text_dropped)
        atomicAdd(&drops->runes, (unsigned long long)(sink.dropped - slot->text_dropped));
    slot->text_dropped = sink.dropped;
}

// Receiver.SetSignalDebounce on the band's current listeners (rx/receiver.go:238-244)
__global__ void k_set_debounce(ListenerSlot *slots, int n, int threshold)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && slots[i].active)
        slots[i].deb.threshold = threshold;
}

hipError_t launch_listen_gather(const float *tap, const float *psd, const sdr_frame_rec *recs, const ListenerSlot *slots, const void *db_tab,
                                uint64_t *raw_bits, float *tr_values, uint8_t *tr_raw, const BatchCursor *cur, ListenGeom g, int n_frames,
                                int n_slots, int n_bands, hipStream_t stream)
{
    const double inv_n2 = 1.0 / ((double)g.n * (double)g.n);
    launch_kernel(k_listen_gather, dim3(((n_frames + 63) / 64 + GATHER_WAVES - 1) / GATHER_WAVES, (n_slots + 63) / 64, n_bands),
                       dim3(64 * GATHER_WAVES), 0, stream, tap, psd, recs,
                       slots, db_tab, raw_bits, tr_values, tr_raw, cur, g, n_frames, n_slots, inv_n2);
    return hipGetLastError();
}

hipError_t launch_listen_decode(ListenerSlot *slots, const uint16_t *morse, const uint64_t *raw_bits,
                                uint64_t *deb_bits, uint32_t *text, uint32_t *text_frames, sdr_edge *edges,
                                uint32_t *edge_counts, uint8_t *tr_deb, DropCounters *drops, const BatchCursor *cur, ListenGeom g,
                                int n_frames, int n_bands, int live_hint, hipStream_t stream)
{
    // live_hint: an upper bound of the signals being decoded (the slots in use)
    const int n_total = n_bands * g.max_listeners;
    const int lanes = (live_hint <= 64 || DECODE_LANES_MAX < 2) ? 1 : (live_hint <= 128 || DECODE_LANES_MAX < 4) ? 2 : DECODE_LANES_MAX;
    const dim3 grid((n_total + lanes * DECODE_WAVES - 1) / (lanes * DECODE_WAVES)), block(64 * DECODE_WAVES);
#define SDR_DECODE_ARGS slots, morse, raw_bits, deb_bits, text, text_frames, edges, edge_counts, tr_deb, drops, cur, g, n_frames, n_total
    if (lanes == 1)
        launch_kernel(k_listen_decode<1>, grid, block, 0, stream, SDR_DECODE_ARGS);
    else if (lanes == 2)
        launch_kernel(k_listen_decode<2>, grid, block, 0, stream, SDR_DECODE_ARGS);
    else
        launch_kernel(k_listen_decode<DECODE_LANES_MAX>, grid, block, 0, stream, SDR_DECODE_ARGS);
#undef SDR_DECODE_ARGS
    return hipGetLastError();
}

hipError_t launch_listener_stop(ListenerSlot *slot, const uint16_t *morse, uint32_t *text, uint32_t *text_frames, int text_cap,
                                uint32_t frame, DropCounters *drops, hipStream_t stream)
{
    hipLaunchKernelGGL(k_listener_stop, dim3(1), dim3(64), 0, stream, slot, morse, text, text_frames, text_cap, frame, drops);
    return hipGetLastError();
}

hipError_t launch_set_debounce(ListenerSlot *slots, int n, int threshold, hipStream_t stream)
{
    hipLaunchKernelGGL(k_set_debounce, dim3((n + 63) / 64), dim3(64), 0, stream, slots, n, threshold);
    return hipGetLastError();
}

}  // namespace sdr
