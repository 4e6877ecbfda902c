// k_listen.hip — the per-signal chain: rx/receiver.go:388-402 -> rx/listener.go:142-148 ->
// cw/spectral.go:48-54 (value > threshold) -> dsp/dsp.go:164-182 (debounce) -> cw/decode.go:202-250
// (Morse timing state machine).  Compiled with -ffp-contract=off.
//
// Two kernels:
//   k_listen_gather  data-parallel over (signal, frame): one lane takes 64 consecutive frames of one
//                    signal, reads the psd values the FFT kernel tapped for it, projects them to dB,
//                    compares against each frame's threshold and packs the 64 results into one word.
//   k_listen_decode  the debouncer and the decoder, taken apart by what is serial in them (cw_stages.h): bit
//                    operations per 64-tick word, a short float64 chain per keying edge, everything else per
//                    edge in parallel, the character bookkeeping per edge - 64 listeners to a workgroup.
#include <hip/hip_runtime.h>

#include "../../include/sdrainer_hip.h"
#include "cw_decoder.h"
#include "cw_stages.h"
#include "gomath.h"
#include "sdr_device.h"

namespace sdr {

// One lane per listener slot, one wave per 64 frames: lane l reads tap[f][l] for the wave's frames (the FFT kernel
// wrote the psd value of the slot's bin there, k_fft_psd.hip "The tap"; neighbouring lanes read neighbouring
// words), projects it to dB - the certified table shortcut of gomath.h, the literal Go algorithm where its
// certificate fails - and collects its own 64 comparison results into one word.  Sixteen waves to a workgroup: round
// 2's one-wave workgroups with the literal logarithm (0.031 ms) went to 128 different CUs and kept an FFT workgroup,
// which needs a whole CU, off each of them for that long; these are 8 workgroups for a few microseconds.
#ifndef SDR_GATHER_WAVES
#define SDR_GATHER_WAVES 16
#endif
constexpr int GATHER_WAVES = SDR_GATHER_WAVES;  // 64-frame words per workgroup, one per wave

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

// ---------------------------------------------------------------------------------------------------------------
// k_listen_decode - debouncer and decoder of every listener, by the stages of cw_stages.h.  A workgroup takes
// DEC_GROUP consecutive listener slots:
//   stage 0   a WAVE per listener, a lane per 64-tick word: debounced bits (bit operations per word, three scans over
//             the words by ballot / shuffle), the edge list - to the batch's edge buffer for the host, and every
//             edge's position to a scratch row (the host's buffer may be shorter than the list) - and the debouncer's
//             state behind the batch;
//   then, DEC_ROUND = 32 edges of every listener at a time, as a pipeline of three stages on different waves with one
//   workgroup barrier per step (step t: A works on round t, B on round t - 1, C + D on round t - 2, through LDS):
//   stage A   a LANE per listener, edge after edge, on TWO waves: the gap threshold's chain over the rising edges (wave
//             0) and the mark threshold's over the falling ones (wave 2) - they are independent of each other.  The only
//             chain left in the kernel: 185 clocks per edge of a polarity;
//   stage B   a thread per (listener, edge), helper waves: thresholds (the square roots), classification, the speed
//             term (the division), the abort ticks of the runs;
//   stage C+D half a helper wave per listener, a lane per edge: the current character WITHOUT a chain - every take empties
//             the character, so what an edge writes follows from the round's events as bit masks (the half-wave's ballots)
//             and the character carried into the round (cw_stages.h round_lane) - then the written runes' places by
//             popcounts of those masks, the table lookup, runes and frames into the text buffer; the speed average's one
//             add and one multiply per da is the only loop, over the round's das.
// The kernel is bound by instruction issue, so the waves are dealt over the CU's four SIMDs by weight (see "who does
// what" below).  Measured (tools/build_abl.sh with -DSDR_DEC_CLOCK, config 3, 8192 frames, 530 edges per listener, 17
// rounds): stage 0 18 k clocks, a chain wave 93 k, the busiest helper 120 k: 0.09 ms per batch where one lane per listener
// walking everything took 0.44 (round 3: 0.51); config 2 - sixteen listeners, one workgroup - 40 -> 90 GS/s.  Sixteen
// workgroups decode config 3's 256 listeners.
// ---------------------------------------------------------------------------------------------------------------
#ifndef SDR_DECODE_GROUP
#define SDR_DECODE_GROUP 16
#endif
#ifndef SDR_DECODE_WAVES
#define SDR_DECODE_WAVES 16
#endif
#ifndef SDR_DECODE_ROUND
#define SDR_DECODE_ROUND 32
#endif
constexpr int DEC_GROUP = SDR_DECODE_GROUP;  // listeners per workgroup: lanes of the waves that run the serial stages
constexpr int DEC_WAVES = SDR_DECODE_WAVES;
constexpr int DEC_ROUND = SDR_DECODE_ROUND;  // edges per listener and round
constexpr int DEC_POS_ROWS = DEC_ROUND + 3;  // a round's positions: one edge before it, two behind
static_assert(DEC_GROUP <= 64 && DEC_ROUND % 2 == 0 && DEC_WAVES >= 4, "lanes of a wave; rising / falling steps alternate");

// The io.Writer of a listener's decoder (cw/decode.go:352): runes and, beside each, the bank frame index of the
// Tick that wrote it (the host stamps TextProcessor.Write with that frame's time: rx/text_processor.go:208-209,
// which is what the listener's silence time-out is measured from, rx/listener.go:126-136).
struct TextSink {
    uint32_t *buf;
    uint32_t *frames;
    uint32_t count, cap, dropped;
    __device__ __forceinline__ void put_if(bool valid, uint32_t r, uint32_t frame)
    {
        const bool ok = valid && count < cap;
        if (ok) {
            buf[count] = r;
            frames[count] = frame;
        }
        count += ok ? 1u : 0u;
        dropped += valid && !ok ? 1u : 0u;
    }
};

// what stage 0 leaves for the later stages, per listener of the group
struct DecodeLocal {
    double t0;            // Decoder.ticks before the batch
    double start0;        // the start the first edge's duration is measured from (offStart / onStart as carried)
    double tick_seconds;  // Decoder.tickSeconds
    int first, end;       // the listener's ticks in the batch
    int n_edges;          // -1: no listener in this slot
    int state0;           // the first edge's new state
    int abort_dits;       // Decoder.abortDecodeAfterDits
    uint32_t first_key, first_frame;  // the character the run in front of the first edge took (0: none), for stage D
};
// an edge's 16 bytes of LDS: stage A's output, replaced by stage B's
union EdgeSlot {
    struct {
        double low, high;
    } chain;
    cw::EdgeRec rec;
};
// the character stage between rounds, per listener of the group
struct CharState {
    double wpm;               // Decoder.wpm
    double gap_threshold_in;  // offThreshold.threshold as carried into the batch (judges the run behind a first, falling edge)
    int32_t len;              // Decoder.currentChar: symbols ...
    uint32_t bits;            // ... their das, most recent in bit 0
    int32_t invalid;          // Decoder.currentCharInvalid
    int32_t pend;             // the run behind the next round's first edge, if that is a falling one, aborts ...
    uint32_t pend_at;         // ... this many ticks into it
    int32_t decoding;         // Decoder.decoding
};

__device__ __forceinline__ int shfl_i(int v, int src) { return __shfl(v, src, 64); }

__global__ __launch_bounds__(64 * DEC_WAVES) void k_listen_decode(ListenerSlot *__restrict__ slots, const uint16_t *__restrict__ morse,
                                                                  const uint64_t *__restrict__ raw_bits, uint64_t *__restrict__ deb_bits,
                                                                  uint32_t *__restrict__ text, uint32_t *__restrict__ text_frames,
                                                                  sdr_edge *__restrict__ edges, uint32_t *__restrict__ edge_counts,
                                                                  uint8_t *__restrict__ tr_deb, DropCounters *__restrict__ drops,
                                                                  const BatchCursor *__restrict__ cur, ListenGeom g, int n_frames, int n_total,
                                                                  uint32_t *__restrict__ edge_pos, int pos_stride)
{
    if (cur)
        g.frame_base = cur->frame_base;
    __shared__ DecodeLocal s_loc[DEC_GROUP];
    __shared__ uint32_t s_pos[4][DEC_POS_ROWS][DEC_GROUP + 1];  // [round & 3][i][l]: position of listener l's edge k0 - 1 + i (behind the last one: the span's end)
    __shared__ double s_now[3][DEC_POS_ROWS][DEC_GROUP + 1];    // [round % 3]: ... and the edge's tick (Decoder.ticks behind its increment)
    __shared__ __attribute__((aligned(16))) EdgeSlot s_edge[4][DEC_ROUND][DEC_GROUP];
    __shared__ CharState s_char[DEC_GROUP];
    __shared__ int s_max_edges;
    if (threadIdx.x == 0)
        s_max_edges = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = (int)(threadIdx.x >> 6);
    const int group0 = blockIdx.x * DEC_GROUP;
    const int n_words = (n_frames + 63) >> 6;
#if defined(SDR_DEC_CLOCK)
    unsigned long long ck[4] = {}, ck_last = __builtin_amdgcn_s_memtime();
#define SDR_DEC_TICK(i)                                               \
    do {                                                              \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        ck[i] += now_ - ck_last;                                      \
        ck_last = now_;                                               \
    } while (0)
#else
#define SDR_DEC_TICK(i)
#endif

    // ---- stage 0: wave per listener, lane per word
    for (int li = wave; li < DEC_GROUP; li += DEC_WAVES) {
        const int idx = group0 + li;
        if (idx >= n_total || !slots[idx].active) {  // (the same for every lane of the wave)
            if (lane == 0)
                s_loc[li].n_edges = -1;
            continue;
        }
        ListenerSlot *slot = &slots[idx];
        const cw::Debouncer deb = slot->deb;
        const int last_state = slot->dec.lastState;
        // frames of this batch before the listener's first one (sdr_attach_at; 0 or less for everybody else)
        const int skip = (int)(slot->start_frame - g.frame_base);
        const cw::TickSpan span{min(max(skip, 0), n_frames), n_frames};
        const uint64_t *rw = raw_bits + (size_t)idx * g.bit_words;
        uint64_t *dw = deb_bits + (size_t)idx * g.bit_words;
        sdr_edge *my_edges = edges + (size_t)idx * g.edge_cap;
        uint32_t *my_pos = edge_pos + (size_t)idx * pos_stride;
        const int band = idx / g.max_listeners, l = idx - band * g.max_listeners;
        const bool pass = deb.threshold < 2;  // dsp/dsp.go:165-167: a pass-through that keeps no state
        int last_restart = cw::deb_run_origin(deb, span);
        bool carried = deb.effectiveState != 0;
        uint32_t top_below = 0;  // the word below the chunk's first: upper half of its effective states
        int n_edges = 0;
        const int last = span.end - 1, last_word = last >> 6;  // (span.end > span.first whenever these are used)
        int last_raw = 0, last_eff = 0;
        for (int w0 = 0; w0 < n_words; w0 += 64) {
            const int w = w0 + lane;
            const bool in = w < n_words;
            const uint64_t r = in ? rw[w] : 0ull, below = in && w > 0 ? rw[w - 1] : 0ull;
            uint64_t eff;
            if (pass) {
                eff = r & cw::span_mask(span, w);
            } else {
                const uint64_t ch = cw::deb_changes(deb, r, below, span, w);
                // the latest restart before this word: the nearest lane below that has one, else what the chunks below left
                const uint64_t has_ch = __ballot(ch != 0);
                const uint64_t ch_below = has_ch & cw::low_mask(lane);
                const int my_restart = ch ? 64 * w + cw::top_bit(ch) : 0;
                const int got = shfl_i(my_restart, ch_below ? cw::top_bit(ch_below) : 0);
                const uint64_t q = cw::deb_qualified(ch, span, w, ch_below ? got : last_restart, deb.threshold);
                const cw::DebFill f = cw::deb_fill(r, q);
                // the state carried into this word: the nearest word below with a qualified tick decides it
                const uint64_t has_q = __ballot(q != 0);
                const uint64_t vals = __ballot(q != 0 && ((r >> cw::top_bit(q | 1ull)) & 1ull));
                const uint64_t q_below = has_q & cw::low_mask(lane);
                const bool cin = q_below ? (vals >> cw::top_bit(q_below)) & 1ull : carried;
                eff = cw::deb_effective(f, cin, span, w);
                if (has_ch)
                    last_restart = shfl_i(my_restart, cw::top_bit(has_ch));
                if (has_q)
                    carried = (vals >> cw::top_bit(has_q)) & 1ull;
            }
            if (in)
                dw[w] = eff;
            if (last_word >= w0 && last_word < w0 + 64) {
                last_raw = shfl_i((int)((r >> (last & 63)) & 1ull), last_word - w0);
                last_eff = shfl_i((int)((eff >> (last & 63)) & 1ull), last_word - w0);
            }
            // edges: the word below's top bit, the number of edges in the words below
            const uint32_t eff_hi = (uint32_t)(eff >> 32);
            uint32_t hi_below = (uint32_t)__shfl_up((int)eff_hi, 1, 64);
            if (lane == 0)
                hi_below = top_below;
            top_below = (uint32_t)shfl_i((int)eff_hi, 63);
            uint64_t e = cw::dec_edges(eff, (uint64_t)hi_below << 32, last_state, span, w);
            const int cnt = cw::count_bits(e);
            int incl = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int v = __shfl_up(incl, d, 64);
                if (lane >= d)
                    incl += v;
            }
            int k = n_edges + incl - cnt;
            n_edges += shfl_i(incl, 63);
            while (e) {
                const int j = cw::bottom_bit(e);
                e &= e - 1ull;
                const uint32_t pos = (uint32_t)(64 * w + j);
                my_pos[k] = pos;
                if (k < g.edge_cap)
                    my_edges[k] = sdr_edge{g.frame_base + pos, (uint32_t)((eff >> j) & 1ull)};
                k++;
            }
            if (g.trace && in) {
                const int cnt_w = min(64, n_frames - 64 * w);
                for (int j = 0; j < cnt_w; j++)
                    tr_deb[((size_t)band * g.stride + 64 * w + j) * g.max_listeners + l] = (eff >> j) & 1ull;
            }
        }
        if (lane == 0) {
            if (!pass && span.first < span.end) {
                slot->deb.lastRawState = last_raw;
                slot->deb.stateCount = last - last_restart + 1;
                slot->deb.effectiveState = last_eff;
            }
            edge_counts[idx] = (uint32_t)n_edges;
            // nothing is lost silently: what did not fit is counted bank-wide (rare; the host reads or polls the totals)
            if (n_edges > g.edge_cap)
                atomicAdd(&drops->edges, (unsigned long long)(n_edges - g.edge_cap));
            DecodeLocal &L = s_loc[li];
            const int state0 = last_state ? 0 : 1;
            L.t0 = slot->dec.ticks;
            L.start0 = state0 ? slot->dec.offStart : slot->dec.onStart;
            L.tick_seconds = slot->dec.tickSeconds;
            L.first = span.first;
            L.end = span.end;
            L.n_edges = n_edges;
            L.state0 = state0;
            L.abort_dits = slot->dec.abortDecodeAfterDits;
            L.first_key = L.first_frame = 0;
            atomicMax(&s_max_edges, n_edges);
        }
    }
    __syncthreads();  // (stage 0's global writes of edge positions are read below by other waves of this workgroup)
    SDR_DEC_TICK(0);

    // ---- who does what from here on
    // Waves go to the CU's four SIMDs round-robin.  The kernel is bound by instruction issue, so the stages are dealt over the
    // SIMDs by weight: the gap threshold's chain (wave 0, SIMD 0) and the mark threshold's (wave 2, SIMD 2) - an edge each per
    // 185 clocks; every other wave is a helper: the six that share a SIMD with a chain come first in the helpers' numbering
    // and take stage B's items (the lightest stage), the eight of SIMD 1 and 3 take stage C + D, a pair of listeners each.
    const bool wave_ar = wave == 0, wave_af = wave == 2, wave_a = wave_ar || wave_af;
    constexpr int N_HELPER_WAVES = DEC_WAVES - 2, N_CD_WAVES = DEC_WAVES / 2, N_B_FIRST = N_HELPER_WAVES - N_CD_WAVES;
    const int helper_wave = wave_a ? -1
                            : (wave & 1) ? N_B_FIRST + (wave >> 2) + ((wave & 2) ? DEC_WAVES / 4 : 0)
                                         : (wave >> 2) - 1 + ((wave & 2) ? DEC_WAVES / 4 - 1 : 0);
    const int cd_wave = helper_wave - N_B_FIRST;  // (>= 0: a wave of SIMD 1 or 3)
    const int helper = helper_wave >= 0 ? helper_wave * 64 + lane : -1;
    constexpr int N_HELPERS = N_HELPER_WAVES * 64;
    constexpr int N_PAIRS = (DEC_GROUP + 1) / 2;  // stage C + D: two listeners to a wave, a lane per edge of the round
    constexpr int D_PER_WAVE = (N_PAIRS + N_CD_WAVES - 1) / N_CD_WAVES;
    static_assert(DEC_ROUND == 32, "stage C + D: half a wave per listener");
    // one lane per listener: wave 1 sets the character stage up (the decoder as carried), waves 0 and 2 run the chains
    const bool serial_lane = (wave_a || wave == 1) && lane < DEC_GROUP;
    DecodeLocal loc{};
    bool on = false;
    ListenerSlot *slot = nullptr;
    if (serial_lane) {
        loc = s_loc[lane];
        on = loc.n_edges >= 0 && loc.first < loc.end;
        slot = &slots[min(group0 + lane, n_total - 1)];
    }
    const int shift = loc.state0 ? 0 : 1;  // the round's rising edges are the ones at shift, shift + 2, ... (a round is an even number of edges)
    // waves A: one threshold's chain
    double t_low = 0, t_high = 0, t_last = 0, t_bound = 0, last_now = 0;
    bool t_moved = false, any_mine = false;
    if (on) {
        if (wave_a) {
            const cw::AdaptiveThreshold &th = wave_ar ? slot->dec.offThreshold : slot->dec.onThreshold;
            t_low = th.low;
            t_high = th.high;
            t_last = th.last;
            t_bound = th.upperBound;
        } else {
            // the run in front of the first edge: the decoder as carried; what is left is the character stage's start
            cw::DecoderState dec = slot->dec;
            const double gap_threshold_in = dec.offThreshold.threshold;
            const int p0 = loc.n_edges ? (int)edge_pos[(size_t)(group0 + lane) * pos_stride] : loc.end;
            cw::Emission em{0u, 0u, false};
            cw::decoder_run(dec, p0 - loc.first, g.frame_base + (uint32_t)loc.first, em);
            s_loc[lane].first_key = em.key;
            s_loc[lane].first_frame = em.frame;
            s_char[lane] = CharState{dec.wpm, gap_threshold_in, dec.charLen, dec.charBits, dec.currentCharInvalid, 0, 0u, dec.decoding};
        }
    }
    const int rounds = (s_max_edges + DEC_ROUND - 1) / DEC_ROUND;
    // a round's edge positions, one before and two behind: a helper thread fetches one of them
    auto fetch_position = [&](int round, int it) -> uint32_t {
        const int li = it / DEC_POS_ROWS, i = it - li * DEC_POS_ROWS, k = round * DEC_ROUND - 1 + i;
        const int ne = s_loc[li].n_edges;
        if (ne < 0 || k < 0)
            return 0u;
        return k < ne ? edge_pos[(size_t)(group0 + li) * pos_stride + k] : (uint32_t)s_loc[li].end;
    };
    // ... and lays it down as a position (stages B, C) and as the edge's tick, Decoder.ticks behind its increment (A, B)
    auto store_position = [&](int round, int it, uint32_t p) {
        const int li = it / DEC_POS_ROWS, i = it - li * DEC_POS_ROWS;
        s_pos[round & 3][i][li] = p;
        s_now[round % 3][i][li] = s_loc[li].t0 + (double)((int)p - s_loc[li].first + 1);
    };
    constexpr int N_POS = DEC_POS_ROWS * DEC_GROUP;
    constexpr int POS_PER_HELPER = (N_POS + N_HELPERS - 1) / N_HELPERS;
    if (helper >= 0 && rounds > 0)
        for (int n = 0; n < POS_PER_HELPER; n++)
            if (helper + n * N_HELPERS < N_POS)
                store_position(0, helper + n * N_HELPERS, fetch_position(0, helper + n * N_HELPERS));
    __syncthreads();
    SDR_DEC_TICK(1);
    // stage D's state: where the next rune of this lane's listener goes (the same in the 32 lanes of a listener)
    uint32_t text_at[D_PER_WAVE] = {}, text_cap_hit[D_PER_WAVE] = {};
    auto d_listener = [&](int n) {  // the listener this lane works for in its wave's n-th pair, -1: none
        const int li = (cd_wave + n * N_CD_WAVES) * 2 + (lane >> 5);
        return cd_wave >= 0 && li < DEC_GROUP && s_loc[li].n_edges >= 0 && s_loc[li].first < s_loc[li].end ? li : -1;
    };
#pragma unroll
    for (int n = 0; n < D_PER_WAVE; n++) {
        const int li = d_listener(n);
        if (li < 0)
            continue;
        const int idx = group0 + li;
        text_at[n] = slots[idx].text_count;
        if (s_loc[li].first_key) {  // the character the run in front of the first edge took
            if (text_at[n] < (uint32_t)g.text_cap) {
                if ((lane & 31) == 0) {
                    text[(size_t)idx * g.text_cap + text_at[n]] = cw::key_to_rune(s_loc[li].first_key, morse);
                    text_frames[(size_t)idx * g.text_cap + text_at[n]] = s_loc[li].first_frame;
                }
                text_at[n]++;
            } else {
                text_cap_hit[n]++;
            }
        }
    }

    for (int t = 0; t < rounds + 2; t++) {
        uint32_t ahead[POS_PER_HELPER];
        const bool fetch = helper >= 0 && t + 1 < rounds;
        if (fetch)
#pragma unroll
            for (int n = 0; n < POS_PER_HELPER; n++)
                if (helper + n * N_HELPERS < N_POS)
                    ahead[n] = fetch_position(t + 1, helper + n * N_HELPERS);
        if (wave_a) {
            // ---- stage A, round t: this wave's threshold, the edges of its polarity
            const int mine = on && t < rounds ? max(0, min(DEC_ROUND, loc.n_edges - t * DEC_ROUND)) : 0;
            const auto &T = s_now[t % 3];
            auto &E = s_edge[t & 3];
            if (__ballot(mine > 0)) {
                const int k_first = wave_ar ? shift : 1 - shift;
                double before = T[k_first][lane], now = T[k_first + 1][lane];
#pragma unroll 4
                for (int kk = k_first; kk < DEC_ROUND; kk += 2) {
                    const double t0 = before, t1 = now;
                    before = T[min(kk + 2, DEC_POS_ROWS - 1)][lane];  // (the next step's, on their way while this one computes)
                    now = T[min(kk + 3, DEC_POS_ROWS - 1)][lane];
                    if (kk < mine) {
                        const double duration = t1 - (t * DEC_ROUND + kk ? t0 : loc.start0);
                        t_moved = cw::chain_step(t_low, t_high, t_last, t_bound, duration) || t_moved;
                        last_now = t1;
                        any_mine = true;
                        E[kk][lane].chain.low = t_low;
                        E[kk][lane].chain.high = t_high;
                    }
                }
            }
        } else if (helper >= 0) {
            // ---- stage B, round t - 1
            const int rb = t - 1;
            if (rb >= 0 && rb < rounds) {
                const int k0 = rb * DEC_ROUND;
                const auto &P = s_pos[rb & 3];
                const auto &T = s_now[rb % 3];
                auto &E = s_edge[rb & 3];
                for (int it = helper; it < DEC_ROUND * DEC_GROUP; it += N_HELPERS) {
                    const int kk = it / DEC_GROUP, li = it - kk * DEC_GROUP, k = k0 + kk;
                    const DecodeLocal &L = s_loc[li];
                    if (k >= L.n_edges)
                        continue;
                    const int p = (int)P[kk + 1][li], p1 = (int)P[kk + 2][li], p2 = (int)P[kk + 3][li];
                    const double now = T[kk + 1][li];
                    const double duration = now - (k ? T[kk][li] : L.start0);
                    const double low = E[kk][li].chain.low, high = E[kk][li].chain.high;
                    cw::EdgeRec rec;
                    if ((L.state0 ^ (k & 1)) != 0)
                        rec = cw::classify_rising(duration, low, high, now, p1 - p - 1, T[kk + 2][li], k + 1 < L.n_edges ? p2 - p1 - 1 : -1, L.abort_dits);
                    else
                        rec = cw::classify_falling(L.tick_seconds, duration, low, high);
                    E[kk][li].rec = rec;
                }
            }
            // ---- stage C + D, round t - 2: half a wave per listener, a lane per edge.  The current character without a chain
            // (cw_stages.h round_lane: what an edge writes follows from the round's events as bit masks - this half's ballots -
            // and the character carried into the round); the runes' places by a prefix sum; runes and frames to the text buffer
            const int rd = t - 2;
            if (cd_wave >= 0 && rd >= 0 && rd < rounds) {
                const auto &P = s_pos[rd & 3];
                const auto &E = s_edge[rd & 3];
#pragma unroll
                for (int n = 0; n < D_PER_WAVE; n++) {
                    const int li = d_listener(n), k = lane & 31, half = lane >> 5;
                    const int cnt = li >= 0 ? max(0, min(DEC_ROUND, s_loc[li].n_edges - rd * DEC_ROUND)) : 0;
                    if (!__ballot(cnt > 0))
                        continue;
                    const bool valid = k < cnt;
                    cw::EdgeRec rec{};
                    CharState cs{};
                    uint32_t frame = 0;
                    if (cnt > 0)
                        cs = s_char[li];
                    if (valid) {
                        rec = E[k][li].rec;
                        frame = g.frame_base + P[k + 1][li];
                    }
                    const bool rising = valid && (rec.flags & cw::ER_STATE);
                    // the abort of the run behind a falling edge: the rising edge before it said (the lane below; the round's
                    // first edge: carried - the batch's first: the gap threshold as carried)
                    if (cnt > 0 && rd == 0 && !s_loc[li].state0) {
                        const int p = (int)P[1][li], p1 = (int)P[2][li];
                        uint32_t at = 0;
                        cs.pend = cw::run_aborts(s_loc[li].t0 + (double)(p - s_loc[li].first + 1), p1 - p - 1, cs.gap_threshold_in, s_loc[li].abort_dits, &at);
                        cs.pend_at = at;
                    }
                    // (cross-lane values come from ballots and from LDS, not from lane shuffles: a shuffle is a trip through the
                    // LDS crossbar, and a dozen dependent ones were most of this stage's time)
                    const int sh = 32 * half;
                    const uint32_t next_aborts = (uint32_t)(__ballot(rising && (rec.flags & cw::ER_ABORT_NEXT)) >> sh);
                    int below_abort = k > 0 ? (int)((next_aborts >> (k - 1)) & 1u) : cs.pend;
                    uint32_t below_at = cs.pend_at;
                    if (valid && k > 0 && !rising)
                        below_at = E[k - 1][li].rec.rise.abort_next_at;
                    const bool abort = valid && (rising ? (rec.flags & cw::ER_ABORT) != 0 : below_abort != 0);
                    const uint32_t abort_at = rising ? rec.abort_at : below_at;
                    const bool falling = valid && !rising;
                    cw::RoundMasks m;
                    m.take = (uint32_t)(__ballot(rising && (rec.flags & cw::ER_TAKE)) >> sh);
                    m.abort = (uint32_t)(__ballot(abort) >> sh);
                    m.invalid = (uint32_t)(__ballot(falling && (rec.flags & cw::ER_INVALID)) >> sh);
                    m.symbol = (uint32_t)(__ballot(falling && (rec.flags & cw::ER_SYMBOL)) >> sh);
                    m.da = (uint32_t)(__ballot(falling && (rec.flags & cw::ER_SYMBOL) && (rec.flags & cw::ER_DA)) >> sh);
                    const cw::CharCarry carry{cs.len, cs.bits, cs.invalid};
                    cw::RoundLane r{0u, 0u, 0};
                    if (valid)
                        r = cw::round_lane(k, rising, m, carry);
                    const uint32_t writes = (uint32_t)(__ballot(r.key_edge != 0 || r.key_abort != 0) >> sh);
                    if (valid)
                        cw::round_lane_invalid(k, r, writes, m, carry);
                    const bool space = rising && (rec.flags & cw::ER_SPACE);
                    // the speed average (:291), the one loop left: over the round's das (their terms straight from LDS, eight
                    // requested at a time), every lane of the half alike
                    double wpm = cs.wpm;
                    for (uint32_t das = m.da; __ballot(das != 0);) {
                        double term[8];
                        bool have[8];
#pragma unroll
                        for (int i = 0; i < 8; i++) {
                            have[i] = das != 0;
                            const int j = das ? cw::bottom_bit(das) : 0;
                            das &= das - 1u;  // (0 & anything: stays 0)
                            term[i] = E[j][max(li, 0)].rec.wpm_term;
                        }
#pragma unroll
                        for (int i = 0; i < 8; i++) {
                            const double next = (wpm + term[i]) / 2.0;
                            wpm = have[i] ? next : wpm;
                        }
                    }
                    // D: the runes' places, from three ballots
                    const uint32_t e_mask = (uint32_t)(__ballot(r.key_edge != 0) >> sh), s_mask = (uint32_t)(__ballot(space) >> sh),
                                   a_mask = (uint32_t)(__ballot(r.key_abort != 0) >> sh);
                    if (__ballot((e_mask | s_mask | a_mask) != 0)) {
                        const uint32_t lower = cw::low_mask32(k);
                        const uint32_t total = (uint32_t)(cw::count_bits32(e_mask) + cw::count_bits32(s_mask) + cw::count_bits32(a_mask)), cap = (uint32_t)g.text_cap;
                        uint32_t at = text_at[n] + (uint32_t)(cw::count_bits32(e_mask & lower) + cw::count_bits32(s_mask & lower) + cw::count_bits32(a_mask & lower));
                        if (li >= 0) {
                            uint32_t *buf = text + (size_t)(group0 + li) * g.text_cap, *frm = text_frames + (size_t)(group0 + li) * g.text_cap;
                            if (r.key_edge && at < cap) {
                                buf[at] = cw::key_to_rune(r.key_edge, morse);
                                frm[at] = frame;
                            }
                            at += r.key_edge ? 1u : 0u;
                            if (space && at < cap) {
                                buf[at] = ' ';
                                frm[at] = frame;
                            }
                            at += space ? 1u : 0u;
                            if (r.key_abort && at < cap) {
                                buf[at] = cw::key_to_rune(r.key_abort, morse);
                                frm[at] = frame + 1u + abort_at;
                            }
                        }
                        const uint32_t room = cap - min(text_at[n], cap);
                        text_cap_hit[n] += total > room ? total - room : 0u;
                        text_at[n] += min(total, room);
                    }
                    // the character carried into the next round
                    if (cnt > 0 && k == 0) {
                        const cw::CharCarry out = cw::round_carry(cnt, writes, m, carry);
                        const int last = cnt - 1;
                        const bool last_next_abort = (next_aborts >> last) & 1u;
                        s_char[li] = CharState{wpm, cs.gap_threshold_in, out.len, out.bits, out.invalid, last_next_abort ? 1 : 0,
                                               last_next_abort ? E[last][li].rec.rise.abort_next_at : 0u, ((m.abort >> last) & 1u) ? 0 : 1};
                    }
                }
            }
        }
        if (fetch)
#pragma unroll
            for (int n = 0; n < POS_PER_HELPER; n++)
                if (helper + n * N_HELPERS < N_POS)
                    store_position(t + 1, helper + n * N_HELPERS, ahead[n]);
        SDR_DEC_TICK(2);
        __syncthreads();
        SDR_DEC_TICK(3);
    }

    // ---- the decoder's state and the text buffer's fill behind the batch
    if (on && wave_a) {
        // (nobody else touches this threshold: what the chain does not hold of it - preset, upperBound, a square root that
        // did not move - is read back rather than kept in registers across the rounds)
        cw::AdaptiveThreshold th = wave_ar ? slot->dec.offThreshold : slot->dec.onThreshold;
        th.low = t_low;
        th.high = t_high;
        th.last = t_last;
        if (t_moved)
            cw::at_update(th);  // updateThreshold :413-416
        if (wave_ar) {
            slot->dec.offThreshold = th;
            if (any_mine)
                slot->dec.onStart = last_now;  // :223
            slot->dec.ticks = loc.t0 + (double)(loc.end - loc.first);
            if (loc.n_edges)
                slot->dec.lastState = loc.state0 ^ ((loc.n_edges - 1) & 1);
            // Frame numbers are 32 bits and compared as differences: a listener that has started must not keep a
            // start_frame that falls 2^31 frames behind (config 5 gets there in 100 days and a decode-mode listener has no
            // time-out) - the difference would turn positive and the listener go deaf.  Once a batch has reached the
            // listener's first frame, both marks move along with the batches (nothing per listener depends on the
            // absolute number after that).
            slot->start_frame = slot->tapped_from = g.frame_base + (uint32_t)n_frames;
        } else {
            slot->dec.onThreshold = th;
            if (any_mine)
                slot->dec.offStart = last_now;  // :231
        }
    }
#pragma unroll
    for (int n = 0; n < D_PER_WAVE; n++) {
        const int li = d_listener(n);
        if (li >= 0 && (lane & 31) == 0) {
            ListenerSlot *sl = &slots[group0 + li];
            const CharState cs = s_char[li];
            sl->dec.decoding = cs.decoding;
            sl->dec.currentCharInvalid = cs.invalid;
            sl->dec.charLen = cs.len;
            sl->dec.charBits = cs.bits;
            sl->dec.wpm = cs.wpm;
            sl->text_count = text_at[n];
            if (text_cap_hit[n]) {  // nothing is lost silently: counted per listener and bank-wide
                sl->text_dropped += text_cap_hit[n];
                atomicAdd(&drops->runes, (unsigned long long)text_cap_hit[n]);
            }
        }
    }
#if defined(SDR_DEC_CLOCK)
    if (blockIdx.x == 0 && lane == 0 && wave < 4)  // (tools only: shader cycles of the chain waves (0, 2) and two helpers)
        printf("decode clocks wave %d: stage0 %llu setup %llu work %llu wait %llu rounds %d\n", wave, ck[0], ck[1], ck[2], ck[3], rounds);
#endif
}

// cw.Decoder.stop for one listener (cw/decode.go:352-354)
__global__ void k_listener_stop(ListenerSlot *slot, const uint16_t *morse, uint32_t *text, uint32_t *text_frames, int text_cap,
                                uint32_t frame, DropCounters *drops)
{
    if (threadIdx.x != 0 || !slot->active)
        return;
    cw::DecoderState dec = slot->dec;
    struct StopSink {
        TextSink to;
        uint32_t frame;
        __device__ void put(uint32_t r) { to.put_if(true, r, frame); }
    } sink{TextSink{text, text_frames, slot->text_count, (uint32_t)text_cap, slot->text_dropped}, frame};
    cw::decoder_stop(dec, morse, sink);
    slot->dec = dec;
    slot->text_count = sink.to.count;
    if (sink.to.dropped != slot->text_dropped)
        atomicAdd(&drops->runes, (unsigned long long)(sink.to.dropped - slot->text_dropped));
    slot->text_dropped = sink.to.dropped;
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
                                int n_frames, int n_bands, uint32_t *edge_pos, int pos_stride, hipStream_t stream)
{
    // edge_pos: [band][listener][pos_stride] scratch, pos_stride >= n_frames (an edge per tick at most)
    const int n_total = n_bands * g.max_listeners;
    launch_kernel(k_listen_decode, dim3((n_total + DEC_GROUP - 1) / DEC_GROUP), dim3(64 * DEC_WAVES), 0, stream, slots, morse, raw_bits, deb_bits, text,
                  text_frames, edges, edge_counts, tr_deb, drops, cur, g, n_frames, n_total, edge_pos, pos_stride);
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
