// The decoder loop of k_listen_decode (cw_decoder.h: decoder_run for a run of equal states, decoder_edge_deferred for
// the edge that ends it, one place per iteration that writes runes) against literal Decoder.Tick calls
// (cw/decode.go:202-250, restated in cw::decoder_tick), tick by tick: same runes, same frames, same decoder state at
// the end of every 64-tick word.  Streams: Morse-like keying at random speeds with jitter, glitches of one tick,
// over-long key-downs, long silences (the abort check), characters of more than eight symbols.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../sdrainer_amd/csrc/cw_decoder.h"

static uint64_t rng_state = 0x243F6A8885A308D3ull;
static uint64_t rnd()
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}

struct Rec {
    uint32_t rune, frame;
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
        auto jit = [&](int len) { return len + (kind >= 1 ? (int)(rnd() % 3) - 1 : 0) > 0 ? len + (kind >= 1 ? (int)(rnd() % 3) - 1 : 0) : 1; };
        if (what < 55) {  // a character of 1..10 symbols
            const int n = 1 + (int)(rnd() % (kind == 2 ? 10 : 6));
            for (int i = 0; i < n; i++) {
                push(1, jit((rnd() & 1) ? dit : 3 * dit));
                push(0, jit(dit));
            }
            push(0, jit(2 * dit));
        } else if (what < 70) {
            push(0, jit(7 * dit));  // word gap
        } else if (what < 78) {
            push(0, 12 * dit + (int)(rnd() % (40 * dit)));  // long silence: the abort check
        } else if (what < 84) {
            push(1, 8 * dit + (int)(rnd() % (20 * dit)));  // over-long key-down
            push(0, jit(dit));
        } else if (what < 92) {
            push((int)(rnd() & 1), 1);  // glitch
        } else {
            dit = 2 + (int)(rnd() % 10);  // speed change
        }
    }
    s.resize((size_t)ticks);
    return s;
}

int main()
{
    uint16_t table[cw::kMorseTableSize];
    cw::build_morse_table(table);
    long mismatches = 0, runes = 0, streams = 0;
    for (int it = 0; it < 400; it++) {
        const int ticks = 64 * (20 + (int)(rnd() % 200)) - (int)(rnd() % 64);
        const std::vector<uint8_t> s = make_stream(ticks, it % 3);
        cw::DecoderState ref, dev;
        cw::decoder_init(ref, 96000, 1024 << (it % 3));
        dev = ref;
        RefSink rs;
        std::vector<Rec> got;
        for (int f0 = 0; f0 < ticks; f0 += 64) {
            const int cnt = ticks - f0 < 64 ? ticks - f0 : 64;
            for (int j = 0; j < cnt; j++) {
                rs.frame = (uint32_t)(f0 + j);
                cw::decoder_tick(ref, s[(size_t)(f0 + j)] != 0, table, rs);
            }
            // the kernel's walk over the word
            int pos = 0;
            while (pos < cnt) {
                const bool cur = dev.lastState != 0;
                int run = 0;
                while (pos + run < cnt && (s[(size_t)(f0 + pos + run)] != 0) == cur)
                    run++;
                cw::Emission em{0u, 0u, false};
                cw::decoder_run(dev, run, (uint32_t)(f0 + pos), em);
                pos += run;
                uint32_t edge_frame = 0;
                if (pos < cnt) {
                    edge_frame = (uint32_t)(f0 + pos);
                    cw::decoder_edge_deferred(dev, !cur, edge_frame, em);
                    pos++;
                }
                if (em.key) {
                    uint32_t r = cw::kUnknownCharacter;
                    if (em.key != cw::kInvalidChar && table[em.key])
                        r = table[em.key];
                    got.push_back({r, em.frame});
                }
                if (em.space)
                    got.push_back({(uint32_t)' ', edge_frame});
            }
            if (memcmp(&ref, &dev, sizeof ref) != 0) {
                if (mismatches < 5)
                    printf("stream %d: decoder state differs after word %d\n", it, f0 / 64);
                mismatches++;
                dev = ref;
            }
        }
        if (got.size() != rs.out.size())
            mismatches++;
        for (size_t i = 0; i < got.size() && i < rs.out.size(); i++)
            if (got[i].rune != rs.out[i].rune || got[i].frame != rs.out[i].frame) {
                if (mismatches < 5)
                    printf("stream %d rune %zu: want %u @%u got %u @%u\n", it, i, rs.out[i].rune, rs.out[i].frame, got[i].rune, got[i].frame);
                mismatches++;
                break;
            }
        runes += (long)rs.out.size();
        streams++;
    }
    printf("%ld streams, %ld runes, mismatches %ld\n", streams, runes, mismatches);
    return mismatches ? 1 : 0;
}
