// The staged listener chain of k_listen_decode (cw_stages.h: debouncer per 64-tick word, edge list, threshold chain per
// edge, classification per edge, character assembly per edge) against literal dsp.BoolDebouncer.Debounce
// (dsp/dsp.go:164-182) + cw.Decoder.Tick (cw/decode.go:202-250) calls, tick by tick, over streams cut into batches of
// random length: same debounced bits, same edges, same runes with the same frames, and the same debouncer and decoder
// state (memcmp) at the end of every batch.  The per-word / per-edge functions are the ones the kernel runs; the scans over
// the words, which the kernel does with a wave's lanes, are plain loops here.
// Streams: Morse-like keying at random speeds with jitter, glitches, over-long marks, long silences (the abort check),
// characters of more than eight symbols, plain noise; debounce thresholds 1 (pass-through), 2, 3, 5, 9, 70 and 200; the first batch of
// every third stream starts inside the batch (sdr_attach_at's `first` > 0).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../sdrainer_amd/csrc/cw_stages.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd()
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}

struct Rec {
    uint32_t rune, frame;
    bool operator!=(const Rec &o) const { return rune != o.rune || frame != o.frame; }
};
struct Edge {
    uint32_t frame, state;
    bool operator!=(const Edge &o) const { return frame != o.frame || state != o.state; }
};
struct RefSink {
    std::vector<Rec> out;
    uint32_t frame = 0;
    void at_run_tick(int) {}
    void put(uint32_t r) { out.push_back({r, frame}); }
};

static std::vector<uint8_t> make_stream(int ticks, int kind)
{
    std::vector<uint8_t> s;
    int dit = 2 + (int)(rnd() % 10);
    while ((int)s.size() < ticks) {
        const int what = (int)(rnd() % 100);
        auto push = [&](int v, int len) {
            for (int i = 0; i < len; i++)
                s.push_back((uint8_t)v);
        };
        if (kind == 3) {  // noise: short runs of either state, now and then a long one
            push((int)(rnd() & 1), (rnd() % 50) ? 1 + (int)(rnd() % 6) : 1 + (int)(rnd() % 300));
            continue;
        }
        auto jit = [&](int len) {
            const int j = len + (kind >= 1 ? (int)(rnd() % 3) - 1 : 0);
            return j > 0 ? j : 1;
        };
        if (what < 55) {
            const int n = 1 + (int)(rnd() % (kind == 2 ? 10 : 6));
            for (int i = 0; i < n; i++) {
                push(1, jit((rnd() & 1) ? dit : 3 * dit));
                push(0, jit(dit));
            }
            push(0, jit(2 * dit));
        } else if (what < 70) {
            push(0, jit(7 * dit));
        } else if (what < 78) {
            push(0, 12 * dit + (int)(rnd() % (40 * dit)));
        } else if (what < 84) {
            push(1, 8 * dit + (int)(rnd() % (20 * dit)));
            push(0, jit(dit));
        } else if (what < 92) {
            const int n = 1 + (int)(rnd() % 3);
            for (int i = 0; i < n; i++)
                push((int)(rnd() & 1), 1);  // glitches
        } else {
            dit = 2 + (int)(rnd() % 10);
        }
    }
    s.resize((size_t)ticks);
    return s;
}

// One batch through the stages.  raw: the batch's raw states, one per frame; first: the listener's first frame in it.
static void staged_batch(cw::Debouncer &deb, cw::DecoderState &dec, const std::vector<uint8_t> &raw, int first, uint32_t frame_base,
                         const uint16_t *table, std::vector<uint64_t> &eff_words, std::vector<Edge> &edges, std::vector<Rec> &runes, bool split_chains, bool parallel_chars)
{
    const int n = (int)raw.size(), n_words = (n + 63) / 64;
    const cw::TickSpan span{first < n ? first : n, n};
    std::vector<uint64_t> rw((size_t)n_words, 0);
    for (int i = 0; i < n; i++)
        if (raw[(size_t)i] && i >= first)  // (k_listen_gather clears the bits before the listener's first frame)
            rw[(size_t)(i >> 6)] |= 1ull << (i & 63);
    eff_words.assign((size_t)n_words, 0);
    edges.clear();
    if (span.first >= span.end)
        return;

    // ---- stage 0
    const bool pass = deb.threshold < 2;
    int last_restart = cw::deb_run_origin(deb, span);
    bool carried = deb.effectiveState != 0;
    std::vector<uint32_t> pos;
    {
        uint64_t eff_below = 0;
        for (int w = 0; w < n_words; w++) {
            const uint64_t r = rw[(size_t)w], below = w ? rw[(size_t)w - 1] : 0;
            uint64_t eff;
            if (pass) {
                eff = r & cw::span_mask(span, w);
            } else {
                const uint64_t ch = cw::deb_changes(deb, r, below, span, w);
                const uint64_t q = cw::deb_qualified(ch, span, w, last_restart, deb.threshold);
                const cw::DebFill f = cw::deb_fill(r, q);
                eff = cw::deb_effective(f, carried, span, w);
                if (ch)
                    last_restart = 64 * w + cw::top_bit(ch);
                if (q)
                    carried = (r >> cw::top_bit(q)) & 1ull;
            }
            eff_words[(size_t)w] = eff;
            uint64_t e = cw::dec_edges(eff, eff_below, dec.lastState, span, w);
            while (e) {
                const int j = cw::bottom_bit(e);
                e &= e - 1;
                pos.push_back((uint32_t)(64 * w + j));
                edges.push_back({frame_base + (uint32_t)(64 * w + j), (uint32_t)((eff >> j) & 1ull)});
            }
            eff_below = eff;
        }
        if (!pass) {
            const int last = span.end - 1;
            deb.lastRawState = (int32_t)((rw[(size_t)(last >> 6)] >> (last & 63)) & 1ull);
            deb.stateCount = last - last_restart + 1;
            deb.effectiveState = (int32_t)((eff_words[(size_t)(last >> 6)] >> (last & 63)) & 1ull);
        }
    }
    const int n_edges = (int)pos.size();
    auto emit = [&](uint32_t key, uint32_t frame) { runes.push_back({cw::key_to_rune(key, table), frame}); };

    // ---- the run in front of the first edge: the decoder as carried
    const double t0 = dec.ticks;
    const int state0 = dec.lastState ? 0 : 1;  // the first edge's new state
    const double gap_threshold_in = dec.offThreshold.threshold;
    {
        cw::Emission em{0u, 0u, false};
        cw::decoder_run(dec, (n_edges ? (int)pos[0] : span.end) - span.first, frame_base + (uint32_t)span.first, em);
        if (em.key)
            emit(em.key, em.frame);
    }
    auto now_of = [&](int k) { return t0 + (double)((int)pos[(size_t)k] - span.first + 1); };
    auto run_behind = [&](int k) { return (k + 1 < n_edges ? (int)pos[(size_t)k + 1] : span.end) - (int)pos[(size_t)k] - 1; };

    // ---- stage A: the gap threshold's chain over the rising edges, the mark threshold's over the falling ones (the
    // kernel runs the two on two waves; even streams go through chain_edge instead: one walk over all edges, the same bits)
    cw::Chain chain = cw::chain_load(dec);
    const double start0 = state0 ? chain.off_start : chain.on_start;
    std::vector<double> lows((size_t)n_edges), highs((size_t)n_edges);
    if (split_chains) {
        for (int pol = 1; pol >= 0; pol--)  // rising first, then falling: neither reads what the other writes
            for (int k = 0; k < n_edges; k++) {
                if (((state0 ^ (k & 1)) != 0) != (pol != 0))
                    continue;
                const double duration = now_of(k) - (k ? now_of(k - 1) : start0);
                if (pol) {
                    chain.off_moved = cw::chain_step(chain.off_low, chain.off_high, chain.off_last, chain.off_bound, duration) || chain.off_moved;
                    chain.on_start = now_of(k);
                    lows[(size_t)k] = chain.off_low;
                    highs[(size_t)k] = chain.off_high;
                } else {
                    chain.on_moved = cw::chain_step(chain.on_low, chain.on_high, chain.on_last, chain.on_bound, duration) || chain.on_moved;
                    chain.off_start = now_of(k);
                    lows[(size_t)k] = chain.on_low;
                    highs[(size_t)k] = chain.on_high;
                }
            }
    } else {
        for (int k = 0; k < n_edges; k++)
            cw::chain_edge(chain, (state0 ^ (k & 1)) != 0, now_of(k), &lows[(size_t)k], &highs[(size_t)k]);
    }
    // ---- stage B
    std::vector<cw::EdgeRec> recs((size_t)n_edges);
    for (int k = 0; k < n_edges; k++) {
        const bool state = (state0 ^ (k & 1)) != 0;
        const double duration = now_of(k) - (k ? now_of(k - 1) : start0);
        if (state)
            recs[(size_t)k] = cw::classify_rising(duration, lows[(size_t)k], highs[(size_t)k], now_of(k), run_behind(k),
                                                  k + 1 < n_edges ? now_of(k + 1) : 0.0, k + 1 < n_edges ? run_behind(k + 1) : -1,
                                                  dec.abortDecodeAfterDits);
        else
            recs[(size_t)k] = cw::classify_falling(dec.tickSeconds, duration, lows[(size_t)k], highs[(size_t)k]);
    }
    // ---- stage C: edge after edge (assemble_edge), or - every other stream - round by round without a chain (round_lane)
    bool pend = false;
    uint32_t pend_at = 0;
    if (n_edges && !state0)  // the batch's first edge is a falling one: the run behind it is judged by the carried gap threshold
        pend = cw::run_aborts(now_of(0), run_behind(0), gap_threshold_in, dec.abortDecodeAfterDits, &pend_at);
    if (!parallel_chars) {
        for (int k = 0; k < n_edges; k++) {
            const cw::EdgeRec &r = recs[(size_t)k];
            const bool rising = r.flags & cw::ER_STATE;
            const uint32_t frame = frame_base + pos[(size_t)k], at = rising ? r.abort_at : pend_at;
            uint32_t key_edge;
            bool space;
            cw::assemble_edge(dec, rising, r, &key_edge, &space);
            const uint32_t key_abort = cw::assemble_abort(dec, rising ? (r.flags & cw::ER_ABORT) != 0 : pend);
            const cw::EdgeEvents ev = cw::edge_events(key_edge, space, key_abort, frame, at);
            if (ev.keys & 0xFFFFu)
                emit(ev.keys & 0xFFFFu, ev.frame);
            if (ev.space)
                emit(cw::kSpaceKey, ev.frame);
            if (ev.keys >> 16)
                emit(ev.keys >> 16, ev.abort_frame);
            if (rising) {
                pend = r.flags & cw::ER_ABORT_NEXT;
                pend_at = r.rise.abort_next_at;
            }
        }
    } else {
        cw::CharCarry carry{dec.charLen, dec.charBits, dec.currentCharInvalid};
        for (int k0 = 0; k0 < n_edges; k0 += 32) {
            const int cnt = n_edges - k0 < 32 ? n_edges - k0 : 32;
            cw::RoundMasks m{0u, 0u, 0u, 0u, 0u};
            bool rising[32], ab[32];
            uint32_t ab_at[32];
            for (int k = 0; k < cnt; k++) {  // (the kernel: a lane per edge, ballots)
                const cw::EdgeRec &r = recs[(size_t)(k0 + k)];
                rising[k] = r.flags & cw::ER_STATE;
                if (rising[k]) {
                    ab[k] = r.flags & cw::ER_ABORT;
                    ab_at[k] = r.abort_at;
                } else {  // the rising edge before it said (the lane below; the round's first edge: carried)
                    ab[k] = pend;
                    ab_at[k] = pend_at;
                }
                if (rising[k]) {
                    pend = r.flags & cw::ER_ABORT_NEXT;
                    pend_at = r.rise.abort_next_at;
                }
                m.take |= (uint32_t)(rising[k] && (r.flags & cw::ER_TAKE)) << k;
                m.abort |= (uint32_t)ab[k] << k;
                m.invalid |= (uint32_t)(!rising[k] && (r.flags & cw::ER_INVALID)) << k;
                m.symbol |= (uint32_t)(!rising[k] && (r.flags & cw::ER_SYMBOL)) << k;
                m.da |= (uint32_t)(!rising[k] && (r.flags & cw::ER_SYMBOL) && (r.flags & cw::ER_DA)) << k;
            }
            cw::RoundLane lanes[32];
            uint32_t writes = 0;
            for (int k = 0; k < cnt; k++) {
                lanes[k] = cw::round_lane(k, rising[k], m, carry);
                writes |= (uint32_t)(lanes[k].key_edge != 0 || lanes[k].key_abort != 0) << k;
            }
            for (int k = 0; k < cnt; k++) {
                cw::round_lane_invalid(k, lanes[k], writes, m, carry);
                const cw::EdgeRec &r = recs[(size_t)(k0 + k)];
                const uint32_t frame = frame_base + pos[(size_t)(k0 + k)];
                if (lanes[k].key_edge)
                    emit(lanes[k].key_edge, frame);
                if (rising[k] && (r.flags & cw::ER_SPACE))
                    emit(cw::kSpaceKey, frame);
                if (lanes[k].key_abort)
                    emit(lanes[k].key_abort, frame + 1u + ab_at[k]);
                if ((m.da >> k) & 1u)  // the speed average (:291): the one loop left
                    dec.wpm = (dec.wpm + r.wpm_term) / 2.0;
            }
            carry = cw::round_carry(cnt, writes, m, carry);
            dec.decoding = ab[cnt - 1] ? 0 : 1;
        }
        dec.charLen = carry.len;
        dec.charBits = carry.bits;
        dec.currentCharInvalid = carry.invalid;
    }
    cw::chain_store(chain, dec);
    dec.ticks = t0 + (double)(span.end - span.first);
    if (n_edges)
        dec.lastState = state0 ^ ((n_edges - 1) & 1);
}

int main()
{
    uint16_t table[cw::kMorseTableSize];
    cw::build_morse_table(table);
    static const int thresholds[] = {1, 2, 3, 5, 9, 70, 200, 3, 3, 2};
    long mismatches = 0, runes_total = 0, edges_total = 0, batches = 0;
    for (int it = 0; it < 600; it++) {
        const int ticks = 3000 + (int)(rnd() % 20000);
        const std::vector<uint8_t> s = make_stream(ticks, it % 4);
        const int thr = thresholds[it % 10];
        cw::Debouncer rdeb, sdeb;
        cw::debouncer_init(rdeb, thr);
        sdeb = rdeb;
        cw::DecoderState rdec, sdec;
        cw::decoder_init(rdec, 96000, 1024 << (it % 3));
        sdec = rdec;
        int at = 0;
        bool first_batch = true;
        while (at < ticks) {
            int n = 1 + (int)(rnd() % ((it & 1) ? 700 : 9000));
            if (n > ticks - at)
                n = ticks - at;
            int first = 0;
            if (first_batch && it % 3 == 0)
                first = (int)(rnd() % (n + 1));  // (n: the listener's first frame lies behind this batch)
            first_batch = false;
            const std::vector<uint8_t> raw(s.begin() + at, s.begin() + at + n);
            // the reference
            RefSink rs;
            std::vector<uint64_t> ref_eff((size_t)(n + 63) / 64, 0);
            std::vector<Edge> ref_edges;
            for (int i = first; i < n; i++) {
                const bool e = cw::debounce(rdeb, raw[(size_t)i] != 0);
                if (e)
                    ref_eff[(size_t)(i >> 6)] |= 1ull << (i & 63);
                if ((int32_t)e != rdec.lastState)
                    ref_edges.push_back({(uint32_t)(at + i), e ? 1u : 0u});
                rs.frame = (uint32_t)(at + i);
                cw::decoder_tick(rdec, e, table, rs);
            }
            // the stages
            std::vector<uint64_t> eff;
            std::vector<Edge> edges;
            std::vector<Rec> runes;
            staged_batch(sdeb, sdec, raw, first, (uint32_t)at, table, eff, edges, runes, ((it >> 2) & 1) != 0, ((it >> 3) & 1) != 0);
            bool bad = false;
            if (eff != ref_eff) {
                if (mismatches < 8)
                    printf("stream %d (threshold %d) batch at %d (+%d, first %d): debounced bits differ\n", it, thr, at, n, first);
                bad = true;
            }
            if (edges.size() != ref_edges.size())
                bad = true;
            for (size_t i = 0; !bad && i < edges.size(); i++)
                if (edges[i] != ref_edges[i])
                    bad = true;
            if (runes.size() != rs.out.size()) {
                if (mismatches < 8)
                    printf("stream %d batch at %d: %zu runes, want %zu\n", it, at, runes.size(), rs.out.size());
                bad = true;
            }
            for (size_t i = 0; !bad && i < runes.size(); i++)
                if (runes[i] != rs.out[i]) {
                    if (mismatches < 8)
                        printf("stream %d batch at %d rune %zu: want %u @%u got %u @%u\n", it, at, i, rs.out[i].rune, rs.out[i].frame, runes[i].rune,
                               runes[i].frame);
                    bad = true;
                }
            if (memcmp(&rdeb, &sdeb, sizeof rdeb) != 0) {
                if (mismatches < 8)
                    printf("stream %d (threshold %d) batch at %d (+%d, first %d): debouncer state differs (%d %d %d | %d %d %d)\n", it, thr, at, n, first,
                           rdeb.effectiveState, rdeb.lastRawState, rdeb.stateCount, sdeb.effectiveState, sdeb.lastRawState, sdeb.stateCount);
                bad = true;
            }
            if (memcmp(&rdec, &sdec, sizeof rdec) != 0) {
                if (mismatches < 8)
                    printf("stream %d batch at %d (+%d, first %d): decoder state differs\n", it, at, n, first);
                bad = true;
            }
            if (bad) {
                mismatches++;
                sdeb = rdeb;
                sdec = rdec;
            }
            runes_total += (long)rs.out.size();
            edges_total += (long)ref_edges.size();
            batches++;
            at += n;
        }
    }
    printf("%ld batches, %ld edges, %ld runes, mismatches %ld\n", batches, edges_total, runes_total, mismatches);
    return mismatches ? 1 : 0;
}
