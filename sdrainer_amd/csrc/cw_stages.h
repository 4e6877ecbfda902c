// cw_stages.h — a listener's chain (dsp.BoolDebouncer, dsp/dsp.go:164-182 -> cw.Decoder.Tick, cw/decode.go:202-250),
// taken apart by what really is serial in it.  One implementation for host and device: k_listen.hip runs these
// functions with a wave's lanes where this file says "per word" / "per edge", tests/emu/emu_stages.cpp runs them in plain
// loops against literal Debounce and Tick calls, tick by tick.
//
// What the reference does per tick, and what it depends on:
//   * Debounce: the effective state becomes the raw state once the raw state has held for `threshold` ticks.  Whether tick
//     t has reached the threshold is a question about the raw bits of the `threshold` ticks before it - independent per
//     tick: bit operations on a 64-tick word (stage 0, "per word"), with three small scans over the words (the position
//     of the last change, the state carried into the word, the number of edges before it).
//   * Tick between edges only counts; the one thing that can happen is the abort check (:244-249), once per run, at a tick
//     that follows from the run's start and the gap threshold (cw_decoder.h decoder_run).
//   * Tick at an edge feeds the run's duration into one AdaptiveThreshold (:392-411): low, high, last <- f(low, high,
//     last, duration).  THAT is the serial chain - five dependent float64 operations per edge of one polarity (stage A,
//     chain_edge).  Everything else an edge does READS the chain and never writes it: the threshold itself (a square
//     root of low * high: a function of the chain's values, recomputed by the reference on every Put that changes them),
//     the dit / da / gap classification, the speed estimate's term (a division), the abort tick of the run behind the
//     edge.  Stage B computes those for every edge independently (classify_rising / classify_falling).
//   * What is left is the bookkeeping of the current character - append a symbol, take the character, write a rune -
//     and the speed average.  The bookkeeping is not a chain either: every take empties the character whatever it
//     held, so what an edge writes is a question about the events since the last take before it, and stage B has said
//     which events happen - stage C answers it per edge with bit counting over a round's events (round_lane; the
//     edge-after-edge form, assemble_edge, is kept: the CPU emulation runs both).  The speed average is one add and one
//     multiply per da, in order: a loop over the round's das.
// Before this split one lane walked all of it per edge: ~1 900 clocks per edge at config 3, ~530 edges per listener and
// 8192-frame batch, 0.44 ms with 64 waves resident for that long.
#pragma once
#include "cw_decoder.h"

namespace cw {

SDR_HD inline uint64_t low_mask(long n) { return n <= 0 ? 0ull : n >= 64 ? ~0ull : ((1ull << n) - 1ull); }
SDR_HD inline int top_bit(uint64_t x)  // x != 0
{
#if defined(__HIP_DEVICE_COMPILE__)
    return 63 - __clzll((long long)x);
#else
    return 63 - __builtin_clzll(x);
#endif
}
SDR_HD inline int bottom_bit(uint64_t x)  // x != 0
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffsll((long long)x) - 1;
#else
    return __builtin_ctzll(x);
#endif
}
SDR_HD inline int count_bits(uint64_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(x);
#else
    return __builtin_popcountll(x);
#endif
}

// The listener's ticks within a batch: positions [first, end) of the batch's frames (first > 0 only for a listener
// bound inside the batch, sdr_attach_at).  Word w holds positions 64 w .. 64 w + 63.
struct TickSpan {
    int first, end;
};
SDR_HD inline uint64_t span_mask(TickSpan s, int w) { return low_mask((long)s.end - 64L * w) & ~low_mask((long)s.first - 64L * w); }

// ---------------------------------------------------------------------------------------------------------------
// Stage 0, per word: the debouncer.
// ---------------------------------------------------------------------------------------------------------------
// Bit j: the count restarts at 1 at tick 64 w + j (dsp/dsp.go:168-170: the raw state differs from the tick before).  The
// tick before the listener's first one is the debouncer's lastRawState; a carried count of 0 restarts either way.
SDR_HD inline uint64_t deb_changes(const Debouncer &d, uint64_t raw, uint64_t raw_below, TickSpan s, int w)
{
    uint64_t ch = raw ^ ((raw << 1) | (raw_below >> 63));
    if ((s.first >> 6) == w) {
        const int b = s.first & 63;
        const bool differs = (int32_t)((raw >> b) & 1ull) != d.lastRawState;
        ch = (ch & ~(1ull << b)) | ((uint64_t)(differs || d.stateCount == 0) << b);
    }
    return ch & span_mask(s, w);
}
// Where the scan "position of the last restart before this word" starts from: the carried run of stateCount ticks began
// that many ticks before the listener's first one.  (Counts beyond 2^30 are taken as 2^30: only `>= threshold` is ever
// asked of them, and a listener gets there after 50 days of one unbroken raw state.)
SDR_HD inline int deb_run_origin(const Debouncer &d, TickSpan s)
{
    const int c = d.stateCount > (1 << 30) ? (1 << 30) : d.stateCount;
    return s.first - c;
}
// Bit j: the count has reached the threshold at tick 64 w + j (:175): the effective state is the raw state there.  A
// restart at p keeps the ticks p .. p + threshold - 2 below it; `last_restart` is the latest one before this word.
SDR_HD inline uint64_t deb_qualified(uint64_t changes, TickSpan s, int w, int last_restart, int threshold)
{
    const int m = threshold - 1;  // (>= 1; the pass-through of threshold < 2 never gets here)
    uint64_t u = changes;
    for (int done = 1; done < m && done < 64;) {
        const int step = done < m - done ? done : m - done;
        u |= u << step;
        done += step;
    }
    const long carried = (long)m - (64L * w - (long)last_restart);  // ticks of this word the carried run still needs
    return ~(u | low_mask(carried)) & span_mask(s, w);
}
// The effective states of a word from its qualified ticks: out[j] = qualified[j] ? raw[j] : out[j - 1], as a parallel
// prefix.  `set`: the states the word decides by itself; `hold`: the ticks before its first qualified one, which keep
// the state carried into the word.
struct DebFill {
    uint64_t set, hold;
};
SDR_HD inline DebFill deb_fill(uint64_t raw, uint64_t qualified)
{
    uint64_t g = raw & qualified, p = ~qualified;
    for (int k = 1; k < 64; k <<= 1) {
        g |= p & (g << k);
        p &= (p << k) | low_mask(k);
    }
    return DebFill{g, p};
}
SDR_HD inline uint64_t deb_effective(DebFill f, bool carried_state, TickSpan s, int w) { return (f.set | (carried_state ? f.hold : 0ull)) & span_mask(s, w); }

// Bit j: the debounced state at tick 64 w + j differs from the decoder's state before it (cw/decode.go:220): an edge.
// `eff_below`: the word below's effective states; the tick before the listener's first one is Decoder.lastState.
SDR_HD inline uint64_t dec_edges(uint64_t eff, uint64_t eff_below, int last_state, TickSpan s, int w)
{
    uint64_t prev = (eff << 1) | (eff_below >> 63);
    if ((s.first >> 6) == w) {
        const int b = s.first & 63;
        prev = (prev & ~(1ull << b)) | ((uint64_t)(last_state != 0) << b);
    }
    return (eff ^ prev) & span_mask(s, w);
}

// ---------------------------------------------------------------------------------------------------------------
// Stage A, per listener, edge after edge: the chain.
// ---------------------------------------------------------------------------------------------------------------
struct Chain {
    double on_low, on_high, on_last, on_bound;      // Decoder.onThreshold: low, high, last, upperBound
    double off_low, off_high, off_last, off_bound;  // Decoder.offThreshold
    double on_start, off_start;                     // Decoder.onStart, offStart
    bool on_moved, off_moved;                       // low / high have changed in this batch
};
SDR_HD inline Chain chain_load(const DecoderState &d)
{
    return Chain{d.onThreshold.low,  d.onThreshold.high,  d.onThreshold.last,  d.onThreshold.upperBound,
                 d.offThreshold.low, d.offThreshold.high, d.offThreshold.last, d.offThreshold.upperBound,
                 d.onStart,          d.offStart,          false,               false};
}
// AdaptiveThreshold.Put (cw/decode.go:392-411) on one threshold's (low, high, last), as straight-line selects - both
// candidate updates are formed and the one the reference's branches take is chosen; a select returns one of its operands
// unchanged, so every value is the one the branching code computes.  `duration`: the run the edge ends (Put is only
// called for runs of minDitTime and more, :254, :279).  Returns whether low / high changed.
SDR_HD inline bool chain_step(double &lo, double &hi, double &last, double bound, double duration)
{
    const bool gate = duration >= kMinDitTime;
    const double highFactor = 2, avgWeight = 0.75, currentWeight = 1.0 - avgWeight;
    const bool use = gate && !(duration >= lo * bound);
    const bool down = last >= duration * highFactor;         // this one shorter: it is the new low sample
    const bool up = !down && duration >= last * highFactor;  // this one longer: the new high sample
    const bool moved = use && (down || up);
    const double lo_sample = down ? duration : last, hi_sample = down ? last : duration;
    const double new_lo = avgWeight * lo + currentWeight * lo_sample;
    const double new_hi = avgWeight * hi + currentWeight * hi_sample;
    lo = moved ? new_lo : lo;
    hi = moved ? new_hi : hi;
    last = use ? duration : last;
    return moved;
}
// The tick `now` (Decoder.ticks after its increment) at which the debounced state changes to `state`: :222-240.  A rising
// edge feeds the gap (off) threshold with the gap it ends, a falling edge the mark (on) threshold with the mark: the two
// thresholds are two independent chains (the kernel runs them on two waves), coupled only through the edges' ticks -
// a run's duration is its edge's tick minus the edge's before.  *low, *high: the fed threshold's values behind the edge.
SDR_HD inline void chain_edge(Chain &c, bool state, double now, double *low, double *high)
{
    const double duration = now - (state ? c.off_start : c.on_start);  // offDuration / onDuration
    if (state) {
        c.on_start = now;
        c.off_moved = chain_step(c.off_low, c.off_high, c.off_last, c.off_bound, duration) || c.off_moved;
        *low = c.off_low;
        *high = c.off_high;
    } else {
        c.off_start = now;
        c.on_moved = chain_step(c.on_low, c.on_high, c.on_last, c.on_bound, duration) || c.on_moved;
        *low = c.on_low;
        *high = c.on_high;
    }
}
// the chain back into the decoder at the end of the batch (updateThreshold :413-416 for a threshold that moved; one that
// did not keeps the square root it has)
SDR_HD inline void chain_store(const Chain &c, AdaptiveThreshold &on, AdaptiveThreshold &off, double *on_start, double *off_start)
{
    on.low = c.on_low;
    on.high = c.on_high;
    on.last = c.on_last;
    off.low = c.off_low;
    off.high = c.off_high;
    off.last = c.off_last;
    if (c.on_moved)
        at_update(on);
    if (c.off_moved)
        at_update(off);
    *on_start = c.on_start;
    *off_start = c.off_start;
}
SDR_HD inline void chain_store(const Chain &c, DecoderState &d) { chain_store(c, d.onThreshold, d.offThreshold, &d.onStart, &d.offStart); }

// ---------------------------------------------------------------------------------------------------------------
// Stage B, per edge: everything that reads the chain.
// ---------------------------------------------------------------------------------------------------------------
constexpr uint32_t ER_STATE = 1;       // the edge is a rising one
constexpr uint32_t ER_TAKE = 2;        // rising: a character or word gap ends the current character (:262-273)
constexpr uint32_t ER_SPACE = 4;       // rising: a word gap: ' ' behind the character
constexpr uint32_t ER_INVALID = 8;     // falling: an over-long mark (:287-288)
constexpr uint32_t ER_SYMBOL = 16;     // falling: a symbol is appended
constexpr uint32_t ER_DA = 32;         // falling: ... a da
constexpr uint32_t ER_ABORT = 64;      // the abort check fires in the run behind this edge, `abort_at` ticks into it
constexpr uint32_t ER_ABORT_NEXT = 128;  // rising: ... in the run behind the NEXT edge (a falling one; same gap threshold)
struct EdgeRec {
    uint32_t flags, abort_at;
    union {
        struct {
            uint32_t abort_next_at, pad;
        } rise;
        double wpm_term;  // falling: ditToWPM(onThreshold.low) (:291)
    };
};
static_assert(sizeof(EdgeRec) == 16, "one 16-byte LDS word per edge");

// the abort check of a run of `run` ticks that starts behind the tick `start` with Decoder.decoding set (every run
// behind an edge: :241): decoder_run's arithmetic.  True: it fires, *at ticks into the run.
SDR_HD inline bool run_aborts(double start, int run, double gap_threshold, int abort_after_dits, uint32_t *at)
{
    const double end = start + (double)run;
    const double upperBound = gap_threshold * (double)abort_after_dits;
    const bool fires = end - start > upperBound;
    const double first_now = ::floor(upperBound) + 1.0 + start;
    *at = fires ? (uint32_t)((int)(first_now - start) - 1) : 0u;  // (below `run` when it fires: the conversion is safe)
    return fires;
}
// `low`, `high`: chain_edge's outputs for this edge; run, run_next: lengths of the runs behind this edge and behind the
// next one (run_next < 0: there is no next edge in this batch); now, now_next: the edges' ticks
SDR_HD inline EdgeRec classify_rising(double duration, double low, double high, double now, int run, double now_next, int run_next,
                                      int abort_after_dits)
{
    EdgeRec r;
    const double threshold = ::sqrt(low * high);  // offThreshold.threshold behind this edge, and until the next rising one
    const bool gate = duration >= kMinDitTime;
    const bool word_gap = duration >= 4.5 * low;
    r.flags = ER_STATE;
    if (gate && (word_gap || duration >= threshold))
        r.flags |= ER_TAKE;
    if (gate && word_gap)
        r.flags |= ER_SPACE;
    if (run_aborts(now, run, threshold, abort_after_dits, &r.abort_at))
        r.flags |= ER_ABORT;
    r.rise.abort_next_at = 0;
    r.rise.pad = 0;
    if (run_next >= 0 && run_aborts(now_next, run_next, threshold, abort_after_dits, &r.rise.abort_next_at))
        r.flags |= ER_ABORT_NEXT;
    return r;
}
SDR_HD inline EdgeRec classify_falling(double tick_seconds, double duration, double low, double high)
{
    EdgeRec r;
    const double threshold = ::sqrt(low * high);  // onThreshold.threshold behind this edge
    const bool gate = duration >= kMinDitTime;
    const bool invalid = gate && duration >= 2 * high;
    const bool symbol = gate && !invalid;
    r.flags = (invalid ? ER_INVALID : 0u) | (symbol ? ER_SYMBOL : 0u) | (duration >= threshold ? ER_DA : 0u);
    r.abort_at = 0;
    r.wpm_term = dit_to_wpm(tick_seconds, low);
    return r;
}

// ---------------------------------------------------------------------------------------------------------------
// Stage C, per listener, edge after edge: the current character.
// ---------------------------------------------------------------------------------------------------------------
// What an edge and the run behind it do to the current character, as selects (the listeners of a group walk their edges
// in lockstep: a branch any of them takes is paid by all - so the two rare things, a ninth symbol and an abort, are
// skipped by a vote of the lanes, everything else is straight-line).  `rising`: the edge's polarity (the caller's steps
// alternate, so it is a constant where this is inlined).  Out: the table key of the character taken at the edge's tick
// (0: none; kInvalidChar), and whether a word gap's ' ' follows it.
#if defined(__HIP_DEVICE_COMPILE__)
#define SDR_ANY_LANE(x) (__builtin_amdgcn_ballot_w64(x) != 0ull)  // true in every active lane if true in any
#else
#define SDR_ANY_LANE(x) (x)
#endif
SDR_HD inline uint32_t take_char_if(DecoderState &d, bool cond)  // take_char(d) if cond, nothing otherwise
{
    const bool has = cond && d.charLen != 0;
    const uint32_t key = d.currentCharInvalid ? kInvalidChar : ((1u << d.charLen) | d.charBits);
    d.currentCharInvalid = has ? 0 : d.currentCharInvalid;
    d.charLen = cond ? 0 : d.charLen;
    d.charBits = cond ? 0u : d.charBits;
    return has ? key : 0u;
}
SDR_HD inline void assemble_edge(DecoderState &d, bool rising, const EdgeRec &r, uint32_t *key_edge, bool *space)
{
    if (rising) {  // onRisingEdge :260-274
        *key_edge = take_char_if(d, (r.flags & ER_TAKE) != 0);
        *space = (r.flags & ER_SPACE) != 0;
    } else {  // onFallingEdge :285-297
        d.currentCharInvalid = (r.flags & ER_INVALID) ? 1 : d.currentCharInvalid;
        const bool symbol = r.flags & ER_SYMBOL, da = r.flags & ER_DA;
        const bool ninth = symbol && d.charLen == kMaxSymbolCount;  // appendSymbol :308-310
        *key_edge = 0;
        if (SDR_ANY_LANE(ninth))
            *key_edge = take_char_if(d, ninth);
        *space = false;
        d.charBits = symbol ? ((d.charBits << 1) | (da ? 1u : 0u)) : d.charBits;
        d.charLen += symbol ? 1 : 0;
        const double wpm = (d.wpm + r.wpm_term) / 2.0;
        d.wpm = (symbol && da) ? wpm : d.wpm;
    }
    d.decoding = 1;  // :241
}
// the abort check in the run behind the edge (:244-249): `abort` - for a rising edge its own ER_ABORT, for a falling one the
// ER_ABORT_NEXT of the rising edge before it (the batch's first edge: run_aborts with the gap threshold as carried)
SDR_HD inline uint32_t assemble_abort(DecoderState &d, bool abort)
{
    uint32_t key = 0;
    if (SDR_ANY_LANE(abort)) {
        key = take_char_if(d, abort);
        d.decoding = abort ? 0 : d.decoding;
    }
    return key;
}
// ---------------------------------------------------------------------------------------------------------------
// Stage C without a chain: the current character per ROUND of up to 32 edges, a lane per edge.
// The character bookkeeping looks serial (append a symbol, take the character) but every take empties the character
// whatever it held, so what an edge's character is made of is a question about the events between it and the last take
// before it - and stage B has already said, per edge, which events happen: as bit masks over the round's edges (a wave's
// ballots on the GPU) every lane answers it for its own edge with bit counting:
//   * boundaries = takes (a rising edge's gap) and aborts (the run behind any edge); the symbols behind the last
//     boundary form the segment; with `n` symbols in it so far the current character holds ((n - 1) mod 8) + 1 of them
//     (appendSymbol's ninth-symbol rule, :308-310, restarts the count) - the carried character counts as symbols before
//     the round's first edge while no boundary has been passed;
//   * a boundary writes a character iff its segment is not empty; a symbol that finds eight writes those eight;
//   * the character is invalid iff an over-long mark (:287-288) lies between the last WRITTEN character and this one
//     (a take that finds nothing to write leaves the flag alone, :320-327);
//   * its table key is made of the most recent symbols' da bits - the round's, then the carried ones.
// Only the speed average (one add and one multiply per da, :291) stays a loop, over the round's das.
// (tests/emu/emu_stages.cpp runs both forms - this one and assemble_edge's walk - against literal Tick calls.)
// ---------------------------------------------------------------------------------------------------------------
struct RoundMasks {  // bit k: edge k of the round
    uint32_t take;     // rising edge with ER_TAKE
    uint32_t abort;    // the abort check fires in the run behind the edge
    uint32_t invalid;  // falling edge with ER_INVALID
    uint32_t symbol;   // falling edge with ER_SYMBOL ...
    uint32_t da;       // ... a da
};
struct CharCarry {  // Decoder.currentChar / currentCharInvalid between rounds
    int32_t len;
    uint32_t bits;
    int32_t invalid;
};
SDR_HD inline uint32_t low_mask32(int n) { return n <= 0 ? 0u : n >= 32 ? ~0u : ((1u << n) - 1u); }
SDR_HD inline uint32_t between32(int lo, int hi) { return low_mask32(hi) & ~low_mask32(lo); }  // bits lo .. hi - 1
SDR_HD inline int top_bit32(uint32_t x)  // x != 0
{
#if defined(__HIP_DEVICE_COMPILE__)
    return 31 - __clz((int)x);
#else
    return 31 - __builtin_clz(x);
#endif
}
SDR_HD inline int count_bits32(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __popc(x);
#else
    return __builtin_popcount(x);
#endif
}
SDR_HD inline int chunk_len(int n) { return n == 0 ? 0 : ((n - 1) & 7) + 1; }
// the bits of the `len` most recent symbols: the round's (`symbols`: the segment's, up to the point in question), then
// the carried ones; most recent symbol in bit 0, as charBits has it
SDR_HD inline uint32_t chunk_bits(uint32_t symbols, uint32_t da, int len, const CharCarry &c)
{
    uint32_t bits = 0, r = symbols;
    int got = 0;
    while (got < len && r) {
        const int b = top_bit32(r);
        bits |= ((da >> b) & 1u) << got;
        r &= ~(1u << b);
        got++;
    }
    if (got < len)
        bits |= (c.bits & ((1u << (len - got)) - 1u)) << got;
    return bits;
}
struct RoundLane {  // what edge k contributes
    uint32_t key_edge, key_abort;  // characters written at the edge's tick / at the abort's (0: none); WITHOUT the invalid flag
    int n_after;                   // symbols in the segment behind the edge's own
};
// lane k, first pass: which characters the edge writes (the invalid flag needs everybody's answer first).  One table
// key is formed in straight line - the common one: a rising edge's gap takes the character - the two rare ones (a ninth
// symbol's, an abort's) behind a vote of the lanes.
SDR_HD inline RoundLane round_lane(int k, bool rising, const RoundMasks &m, const CharCarry &c)
{
    const uint32_t boundaries = (m.take | m.abort) & low_mask32(k);
    const int seg_lo = boundaries ? top_bit32(boundaries) + 1 : 0;
    const uint32_t before = m.symbol & between32(seg_lo, k);
    const int n_before = count_bits32(before) + (boundaries ? 0 : c.len);
    const bool take = (m.take >> k) & 1u, abort = (m.abort >> k) & 1u;
    const bool symbol = !rising && ((m.symbol >> k) & 1u);
    RoundLane r{0u, 0u, n_before + (symbol ? 1 : 0)};
    const bool gap_takes = rising && take && n_before > 0;
    if (SDR_ANY_LANE(gap_takes)) {
        const int len = chunk_len(n_before);
        r.key_edge = gap_takes ? ((1u << len) | chunk_bits(before, m.da, len, c)) : 0u;
    }
    const bool ninth = symbol && chunk_len(n_before) == kMaxSymbolCount;  // appendSymbol :308-310
    if (SDR_ANY_LANE(ninth))
        r.key_edge = ninth ? ((1u << kMaxSymbolCount) | chunk_bits(before, m.da, kMaxSymbolCount, c)) : r.key_edge;
    // the abort in the run behind the edge: behind a rising edge whose gap took the character there is nothing left
    const bool aborts = abort && !(rising && take) && r.n_after > 0;
    if (SDR_ANY_LANE(aborts)) {
        const int len = chunk_len(r.n_after);
        r.key_abort = aborts ? ((1u << len) | chunk_bits(before | (symbol ? 1u << k : 0u), m.da, len, c)) : 0u;
    }
    return r;
}
// second pass: `writes` = the edges that write a character (either kind).  The invalid flag of edge k's characters.
SDR_HD inline void round_lane_invalid(int k, RoundLane &r, uint32_t writes, const RoundMasks &m, const CharCarry &c)
{
    const uint32_t earlier = writes & low_mask32(k);
    const int lo = earlier ? top_bit32(earlier) + 1 : 0;
    const bool inv = (m.invalid & between32(lo, k + 1)) != 0u || (!earlier && c.invalid);
    if (r.key_edge) {
        r.key_edge = inv ? kInvalidChar : r.key_edge;
        r.key_abort = r.key_abort;  // (a second character at the same edge: nothing invalid can lie between the two)
    } else if (r.key_abort) {
        r.key_abort = inv ? kInvalidChar : r.key_abort;
    }
}
// the character carried out of a round of `cnt` edges
SDR_HD inline CharCarry round_carry(int cnt, uint32_t writes, const RoundMasks &m, const CharCarry &c)
{
    const uint32_t boundaries = (m.take | m.abort) & low_mask32(cnt);
    const int seg_lo = boundaries ? top_bit32(boundaries) + 1 : 0;
    const uint32_t seg = m.symbol & between32(seg_lo, cnt);
    const int n = count_bits32(seg) + (boundaries ? 0 : c.len);
    CharCarry o;
    o.len = chunk_len(n);
    o.bits = chunk_bits(seg, m.da, o.len, c);
    const uint32_t w = writes & low_mask32(cnt);
    const int lo = w ? top_bit32(w) + 1 : 0;
    o.invalid = ((m.invalid & between32(lo, cnt)) != 0u || (!w && c.invalid)) ? 1 : 0;
    return o;
}

// What an edge's step may write, in the reference's order: the edge's character, the word gap's ' ' (both at the edge's
// frame), the abort's character (at its own).
constexpr uint32_t kSpaceKey = 0xFFFEu;  // ' ' in a list of table keys
struct EdgeEvents {
    uint32_t keys;   // the edge's character | the abort's << 16 (table keys, kInvalidChar; 0: none)
    uint32_t space;  // a word gap's ' ' between them
    uint32_t frame, abort_frame;
};
SDR_HD inline EdgeEvents edge_events(uint32_t key_edge, bool space, uint32_t key_abort, uint32_t frame, uint32_t abort_at)
{
    return EdgeEvents{key_edge | (key_abort << 16), space ? 1u : 0u, frame, frame + 1u + abort_at};
}
// a key as the rune the reference writes (decodeCurrentChar :315-350)
SDR_HD inline uint32_t key_to_rune(uint32_t key, const uint16_t *table)
{
    if (key == kSpaceKey)
        return ' ';
    if (key == kInvalidChar)
        return kUnknownCharacter;
    const uint32_t looked_up = table[key];
    return looked_up ? looked_up : kUnknownCharacter;
}

}  // namespace cw
