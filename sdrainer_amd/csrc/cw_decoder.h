// cw_decoder.h — debouncer + Morse timing state machine, one implementation for host and device.
//
// Mirrors dsp.BoolDebouncer (dsp/dsp.go:139-182), cw.AdaptiveThreshold (cw/decode.go:360-431) and
// cw.Decoder (cw/decode.go:108-354).  State is plain-old-data so a listener's decoder lives in HBM
// between batches and is loaded into registers by the per-signal kernel.  All timing arithmetic is
// float64, as in the reference (`type ticks float64`).
#pragma once
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define SDR_HD __host__ __device__
#else
#define SDR_HD
#endif

namespace cw {

constexpr uint32_t kUnknownCharacter = 0xA6;  // cw/decode.go:33
constexpr int kMaxSymbolCount = 8;            // cw/decode.go:36
constexpr double kMinDitTime = 2.0;           // cw/decode.go:38
constexpr int kDefaultWPM = 20;               // cw/decode.go:35
constexpr int kMorseTableSize = 512;          // code key = (1 << len) | bits, len <= 8

// dsp.BoolDebouncer (dsp/dsp.go:139-146)
struct Debouncer {
    int32_t threshold;
    int32_t effectiveState;
    int32_t lastRawState;
    int32_t stateCount;
};

SDR_HD inline void debouncer_init(Debouncer &d, int threshold)
{
    d.threshold = threshold;
    d.effectiveState = 0;
    d.lastRawState = 0;
    d.stateCount = 0;
}

// dsp/dsp.go:164-182 Debounce
SDR_HD inline bool debounce(Debouncer &d, bool raw)
{
    if (d.threshold < 2)
        return raw;
    if ((int32_t)raw != d.lastRawState)
        d.stateCount = 1;
    else
        d.stateCount++;
    d.lastRawState = raw;
    if (d.stateCount >= d.threshold) {
        if ((int32_t)raw != d.effectiveState)
            d.effectiveState = raw;
    }
    return d.effectiveState != 0;
}

// cw/decode.go:360-369
struct AdaptiveThreshold {
    double preset, upperBound, low, high, last, threshold;
};

SDR_HD inline void at_update(AdaptiveThreshold &t) { t.threshold = ::sqrt(t.low * t.high); }  // :413-416
SDR_HD inline void at_reset(AdaptiveThreshold &t)                                             // :380-385
{
    t.low = t.preset;
    t.high = 3 * t.low;
    t.last = t.low;
    at_update(t);
}
SDR_HD inline void at_new(AdaptiveThreshold &t, double preset)  // :371-378
{
    t.preset = preset;
    t.upperBound = 10;
    at_reset(t);
}
SDR_HD inline void at_preset(AdaptiveThreshold &t, double preset)  // :387-390
{
    t.preset = preset;
    at_reset(t);
}
SDR_HD inline void at_put(AdaptiveThreshold &t, double duration)  // :392-411
{
    const double highFactor = 2;
    const double avgWeight = 0.75;
    const double currentWeight = 1.0 - avgWeight;
    if (duration >= t.low * t.upperBound)
        return;
    bool moved = false;
    if (t.last >= duration * highFactor) {
        t.low = avgWeight * t.low + currentWeight * duration;
        t.high = avgWeight * t.high + currentWeight * t.last;
        moved = true;
    } else if (duration >= t.last * highFactor) {
        t.low = avgWeight * t.low + currentWeight * t.last;
        t.high = avgWeight * t.high + currentWeight * duration;
        moved = true;
    }
    t.last = duration;
    // updateThreshold() recomputes sqrt(low*high) on every Put; when neither moved the value is the one
    // already stored (sqrt is a function), so the ~30-instruction float64 sqrt is skipped
    if (moved)
        at_update(t);
}

// cw.Decoder state (cw/decode.go:108-129).  currentChar is packed: `len` symbols, bit i of `bits`
// (MSB-first order of arrival) is 1 for a Da, 0 for a Dit.
struct DecoderState {
    double tickSeconds;
    double ticks;
    double onStart, offStart;
    double wpm;
    AdaptiveThreshold onThreshold, offThreshold;
    int32_t lastState;
    int32_t decoding;
    int32_t abortDecodeAfterDits;
    int32_t currentCharInvalid;
    int32_t charLen;
    uint32_t charBits;
};

// Where decoded runes go.  `Sink::put(uint32_t rune)` is the io.Writer of the reference.
SDR_HD inline double wpm_to_dit(const DecoderState &d, double wpm)  // :191-195
{
    const double ditSeconds = 60.0 / (50.0 * wpm);
    return ::ceil(ditSeconds / d.tickSeconds);
}
SDR_HD inline double dit_to_wpm(double tickSeconds, double ditTicks)  // :197-200
{
    const double ditSeconds = ditTicks * tickSeconds;
    return 60.0 / (50.0 * ditSeconds);
}
SDR_HD inline double dit_to_wpm(const DecoderState &d, double ditTicks) { return dit_to_wpm(d.tickSeconds, ditTicks); }

SDR_HD inline void decoder_init(DecoderState &d, int sampleRate, int blockSize)  // NewDecoder :131-147
{
    d.tickSeconds = (double)blockSize / (double)sampleRate;
    d.ticks = 0;
    d.onStart = 0;
    d.offStart = 0;
    d.wpm = kDefaultWPM;
    d.lastState = 0;
    d.decoding = 0;
    d.abortDecodeAfterDits = 10;
    d.currentCharInvalid = 0;
    d.charLen = 0;
    d.charBits = 0;
    const double dit = wpm_to_dit(d, d.wpm);
    at_new(d.onThreshold, dit);
    at_new(d.offThreshold, dit);
}

SDR_HD inline void decoder_clear(DecoderState &d)  // Clear :172-178
{
    d.decoding = 0;
    d.charLen = 0;
    d.charBits = 0;
    d.ticks = 0;
    d.onStart = 0;
    d.offStart = 0;
}
SDR_HD inline void decoder_preset_wpm(DecoderState &d, int wpm)  // presetWPM :180-185
{
    d.wpm = (double)wpm;
    const double dit = wpm_to_dit(d, d.wpm);
    at_preset(d.onThreshold, dit);
    at_preset(d.offThreshold, dit);
}
SDR_HD inline void decoder_reset(DecoderState &d)  // Reset :166-170 (lastState, currentCharInvalid survive)
{
    decoder_preset_wpm(d, kDefaultWPM);
    decoder_clear(d);
    at_reset(d.onThreshold);
}

template <class Sink>
SDR_HD inline void decode_current_char(DecoderState &d, const uint16_t *table, Sink &out)  // :315-350
{
    if (d.charLen == 0)
        return;
    if (d.currentCharInvalid) {
        d.currentCharInvalid = 0;
        d.charLen = 0;
        d.charBits = 0;
        out.put(kUnknownCharacter);
        return;
    }
    const uint32_t key = (1u << d.charLen) | d.charBits;
    const uint32_t r = table[key];
    out.put(r ? r : kUnknownCharacter);
    d.charLen = 0;
    d.charBits = 0;
}

template <class Sink>
SDR_HD inline void append_symbol(DecoderState &d, bool da, const uint16_t *table, Sink &out)  // :307-313
{
    if (d.charLen == kMaxSymbolCount)
        decode_current_char(d, table, out);
    d.charBits = (d.charBits << 1) | (da ? 1u : 0u);
    d.charLen++;
}

template <class Sink>
SDR_HD inline void on_rising_edge(DecoderState &d, double offDuration, const uint16_t *table, Sink &out)  // :252-275
{
    if (offDuration < kMinDitTime)
        return;
    at_put(d.offThreshold, offDuration);
    const double threshold = d.offThreshold.threshold;
    const double upperThreshold = 4.5 * d.offThreshold.low;
    if (offDuration >= upperThreshold) {
        decode_current_char(d, table, out);
        out.put(' ');
    } else if (offDuration >= threshold) {
        decode_current_char(d, table, out);
    }
}

template <class Sink>
SDR_HD inline void on_falling_edge(DecoderState &d, double onDuration, const uint16_t *table, Sink &out)  // :277-298
{
    if (onDuration < kMinDitTime)
        return;
    at_put(d.onThreshold, onDuration);
    const double threshold = d.onThreshold.threshold;
    const double upperThreshold = 2 * d.onThreshold.high;
    if (onDuration >= upperThreshold) {
        d.currentCharInvalid = 1;
    } else if (onDuration >= threshold) {
        append_symbol(d, true, table, out);
        d.wpm = (d.wpm + dit_to_wpm(d, d.onThreshold.low)) / 2.0;
    } else {
        append_symbol(d, false, table, out);
    }
}

template <class Sink>
SDR_HD inline void decoder_tick(DecoderState &d, bool state, const uint16_t *table, Sink &out)  // Tick :202-250
{
    d.ticks += 1;
    const double now = d.ticks;
    if ((int32_t)state != d.lastState) {
        if (state) {
            d.onStart = now;
            on_rising_edge(d, now - d.offStart, table, out);
        } else {
            d.offStart = now;
            on_falling_edge(d, now - d.onStart, table, out);
        }
        d.decoding = 1;
    }
    d.lastState = state;
    const double currentDuration = state ? now - d.onStart : now - d.offStart;
    const double upperBound = d.offThreshold.threshold * (double)d.abortDecodeAfterDits;
    if (d.decoding && currentDuration > upperBound) {
        d.decoding = 0;
        decode_current_char(d, table, out);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Pieces of the decoder as k_listen_decode runs it (cw_stages.h has the rest).  decodeCurrentChar (cw/decode.go:315-350)
// is reached from four places in a tick (the abort check, a character gap, a word gap, a ninth symbol); each of them
// empties the current character, so the stages only TAKE the character (a table key, 0 = nothing) and the caller looks
// it up and writes the rune.  (tests/emu/emu_stages.cpp: identical to Decoder.Tick tick by tick.)
constexpr uint32_t kInvalidChar = 0xFFFFu;  // a character with an over-long Da in it: decodes to kUnknownCharacter (table keys are below 512: a key fits 16 bits)

// the state changes of decodeCurrentChar, without the output: returns the table key of the character taken
// (kInvalidChar for an invalid one), 0 if there was none
SDR_HD inline uint32_t take_char(DecoderState &d)
{
    const bool has = d.charLen != 0;
    const uint32_t key = d.currentCharInvalid ? kInvalidChar : ((1u << d.charLen) | d.charBits);
    d.currentCharInvalid = has ? 0 : d.currentCharInvalid;  // (survives while there is no symbol to report it on)
    d.charLen = 0;
    d.charBits = 0;  // (zero already whenever charLen is)
    return has ? key : 0u;
}

struct Emission {
    uint32_t key;    // character to look up and write, 0 = none
    uint32_t frame;  // the frame of the tick that writes it
    bool space;      // a word gap: ' ' after the character, stamped with the edge's frame
};

// `k` consecutive Tick(state) calls with state == lastState, the first of them frame `run_base`, in closed form.  Between
// edges Tick only counts (`ticks++`) and checks `decoding && currentDuration > upperBound` (:244-249); neither threshold
// changes, currentDuration = now - start is an exact integer, so the check first fires at the tick where
// now == floor(upperBound) + 1 + start (if that tick is within the run) and never again (it clears `decoding`).
SDR_HD inline void decoder_run(DecoderState &d, int k, uint32_t run_base, Emission &em)
{
    const double end = d.ticks + (double)k;
    const double start = d.lastState ? d.onStart : d.offStart;
    const double upperBound = d.offThreshold.threshold * (double)d.abortDecodeAfterDits;
    if (d.decoding && end - start > upperBound) {  // (rare)
        const double first_now = ::floor(upperBound) + 1.0 + start;
        d.decoding = 0;
        const uint32_t key = take_char(d);
        if (key) {
            em.key = key;
            em.frame = run_base + (uint32_t)((int)(first_now - d.ticks) - 1);
        }
    }
    d.ticks = end;
}

template <class Sink>
SDR_HD inline void decoder_stop(DecoderState &d, const uint16_t *table, Sink &out)  // stop :352-354
{
    decode_current_char(d, table, out);
}

// (host only)
// Morse code table: code string -> rune.  The reference imports it from
// github.com/ftl/digimodes v0.0.0-20231231131023-cffadad68e9e (cw.Code), which is not vendored;
// this is the published International Morse code (ITU-R M.1677-1) plus the entries the reference's
// tests pin ('ä' .-.-, '§' = 8 dits: cw/decode_test.go:23-29,184-192).
struct MorseEntry {
    uint32_t rune;
    const char *code;
};
inline const MorseEntry *morse_entries(int *count)
{
    static const MorseEntry E[] = {
        {'a', ".-"},     {'b', "-..."},   {'c', "-.-."},   {'d', "-.."},     {'e', "."},      {'f', "..-."},
        {'g', "--."},    {'h', "...."},   {'i', ".."},     {'j', ".---"},    {'k', "-.-"},    {'l', ".-.."},
        {'m', "--"},     {'n', "-."},     {'o', "---"},    {'p', ".--."},    {'q', "--.-"},   {'r', ".-."},
        {'s', "..."},    {'t', "-"},      {'u', "..-"},    {'v', "...-"},    {'w', ".--"},    {'x', "-..-"},
        {'y', "-.--"},   {'z', "--.."},   {'0', "-----"},  {'1', ".----"},   {'2', "..---"},  {'3', "...--"},
        {'4', "....-"},  {'5', "....."},  {'6', "-...."},  {'7', "--..."},   {'8', "---.."},  {'9', "----."},
        {'.', ".-.-.-"}, {',', "--..--"}, {'?', "..--.."}, {'/', "-..-."},   {'=', "-...-"},  {'+', ".-.-."},
        {'-', "-....-"}, {'@', ".--.-."}, {':', "---..."}, {';', "-.-.-."},  {'\'', ".----."}, {'"', ".-..-."},
        {'(', "-.--."},  {')', "-.--.-"}, {'_', "..--.-"}, {'!', "-.-.--"},  {'&', ".-..."},  {'$', "...-..-"},
        {0xE4, ".-.-"},  {0xF6, "---."},  {0xFC, "..--"},  {0xA7, "........"},
    };
    *count = (int)(sizeof E / sizeof E[0]);
    return E;
}
// generateDecodeTable (cw/decode.go:149-157) into a flat 512-entry array keyed by (1<<len)|bits
inline void build_morse_table(uint16_t *table)
{
    for (int i = 0; i < kMorseTableSize; i++)
        table[i] = 0;
    int n;
    const MorseEntry *E = morse_entries(&n);
    for (int i = 0; i < n; i++) {
        uint32_t bits = 0;
        int len = 0;
        for (const char *c = E[i].code; *c; c++, len++)
            bits = (bits << 1) | (*c == '-' ? 1u : 0u);
        table[(1u << len) | bits] = (uint16_t)E[i].rune;
    }
}

}  // namespace cw
