// cw::debounce_word (a run of equal raw states at a time) against 64 literal dsp.BoolDebouncer.Debounce calls
// (dsp/dsp.go:164-182, restated in cw::debounce), state carried from word to word, for every threshold 0..9 and
// for words of every length: random words with short and long runs.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../sdrainer_amd/csrc/cw_decoder.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd()
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}

int main()
{
    long mismatches = 0, words = 0;
    for (int threshold = 0; threshold < 10; threshold++) {
        cw::Debouncer a, b;
        cw::debouncer_init(a, threshold);
        cw::debouncer_init(b, threshold);
        for (int it = 0; it < 200000; it++) {
            uint64_t raw;
            switch (it % 4) {
            case 0: raw = rnd(); break;                            // runs of 1-3
            case 1: raw = rnd() & rnd() & rnd(); break;            // mostly off
            case 2: raw = ~(rnd() & rnd() & rnd()); break;         // mostly on
            default: {                                             // long runs
                raw = 0;
                int pos = 0, v = (int)(rnd() & 1);
                while (pos < 64) {
                    const int len = 1 + (int)(rnd() % 40);
                    for (int j = pos; j < pos + len && j < 64; j++)
                        raw |= (uint64_t)v << j;
                    pos += len;
                    v ^= 1;
                }
            }
            }
            if (it % 97 == 0)
                raw = (it & 1) ? ~0ull : 0ull;
            const int cnt = (it % 5 == 0) ? 1 + (int)(rnd() % 64) : 64;
            uint64_t want = 0;
            for (int j = 0; j < cnt; j++)
                want |= (uint64_t)cw::debounce(a, (raw >> j) & 1ull) << j;
            const uint64_t got = cw::debounce_word(b, raw, cnt);
            words++;
            if (got != want || memcmp(&a, &b, sizeof a) != 0) {
                if (mismatches < 5)
                    printf("threshold %d word %d cnt %d raw %016llx want %016llx got %016llx\n", threshold, it, cnt,
                           (unsigned long long)raw, (unsigned long long)want, (unsigned long long)got);
                mismatches++;
                b = a;
            }
        }
    }
    printf("%ld words, mismatches %ld\n", words, mismatches);
    return mismatches ? 1 : 0;
}
