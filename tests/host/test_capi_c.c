/* A PLAIN C caller of the drop-in boundary (include/sdrainer_hip.h), what a cgo shim's C half is.
 *   - compiled with gcc -std=c11 -Wall -Werror -pedantic in the CPU suite: the header is C, not just C++
 *     (tests/test_capi_c.py), and the structs have the layout the ctypes binding asserts;
 *   - run on the GPU as a -m gpu test: create -> push_iq -> process_staged -> attach -> push / process -> poll -> destroy,
 *     printing what it got; the Python test compares the peaks and the decoded text with the oracle's.
 * usage: test_capi_c layout                      print sizeof / offsetof of every struct of the header
 *        test_capi_c run <iq.f32> <rate> <n> <frames> <edge> <bin> [<bin> ...]
 *                                                 frames of n complex samples from the file; the listeners are attached
 *                                                 behind the first 100 frames, as a strainer that has just found them
 */
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/sdrainer_hip.h"

_Static_assert(sizeof(sdr_config) == 56, "sdr_config");
_Static_assert(sizeof(sdr_peak) == 40, "sdr_peak");
_Static_assert(sizeof(sdr_frame_rec) == 40, "sdr_frame_rec");
_Static_assert(sizeof(sdr_edge) == 8, "sdr_edge");
_Static_assert(sizeof(sdr_chunk_result) == 24, "sdr_chunk_result");
_Static_assert(sizeof(sdr_listener_result) == 24, "sdr_listener_result");
_Static_assert(sizeof(sdr_results) == 128, "sdr_results");

#define OFF(type, field) printf(#type "." #field " %zu\n", offsetof(type, field))

static int layout(void)
{
    printf("sizeof sdr_config %zu\n", sizeof(sdr_config));
    OFF(sdr_config, struct_size);
    OFF(sdr_config, n_bands);
    OFF(sdr_config, sample_rate);
    OFF(sdr_config, block_size);
    OFF(sdr_config, edge_width);
    OFF(sdr_config, peak_threshold);
    OFF(sdr_config, signal_debounce);
    OFF(sdr_config, max_listeners);
    OFF(sdr_config, max_batch_frames);
    OFF(sdr_config, max_peaks);
    OFF(sdr_config, find_peaks);
    OFF(sdr_config, trace);
    OFF(sdr_config, device_id);
    printf("sizeof sdr_peak %zu\n", sizeof(sdr_peak));
    OFF(sdr_peak, from);
    OFF(sdr_peak, to);
    OFF(sdr_peak, from_frequency);
    OFF(sdr_peak, to_frequency);
    OFF(sdr_peak, signal_frequency);
    OFF(sdr_peak, signal_value);
    OFF(sdr_peak, signal_bin);
    printf("sizeof sdr_frame_rec %zu\n", sizeof(sdr_frame_rec));
    OFF(sdr_frame_rec, min_mean);
    OFF(sdr_frame_rec, dev_in);
    OFF(sdr_frame_rec, variance);
    OFF(sdr_frame_rec, nf_in);
    OFF(sdr_frame_rec, noise_dev);
    OFF(sdr_frame_rec, noise_floor);
    OFF(sdr_frame_rec, peak_thr);
    OFF(sdr_frame_rec, listen_thr);
    printf("sizeof sdr_edge %zu\n", sizeof(sdr_edge));
    OFF(sdr_edge, frame);
    OFF(sdr_edge, state);
    printf("sizeof sdr_chunk_result %zu\n", sizeof(sdr_chunk_result));
    OFF(sdr_chunk_result, band);
    OFF(sdr_chunk_result, n_peaks);
    OFF(sdr_chunk_result, frame);
    OFF(sdr_chunk_result, first_peak);
    OFF(sdr_chunk_result, peaks_found);
    printf("sizeof sdr_listener_result %zu\n", sizeof(sdr_listener_result));
    OFF(sdr_listener_result, band);
    OFF(sdr_listener_result, listener);
    OFF(sdr_listener_result, first_edge);
    OFF(sdr_listener_result, n_edges);
    OFF(sdr_listener_result, first_rune);
    OFF(sdr_listener_result, n_runes);
    printf("sizeof sdr_results %zu\n", sizeof(sdr_results));
    OFF(sdr_results, struct_size);
    OFF(sdr_results, n_frames);
    OFF(sdr_results, batch_index);
    OFF(sdr_results, first_frame);
    OFF(sdr_results, chunks);
    OFF(sdr_results, chunks_cap);
    OFF(sdr_results, n_chunks);
    OFF(sdr_results, peaks);
    OFF(sdr_results, peaks_cap);
    OFF(sdr_results, n_peaks);
    OFF(sdr_results, listeners);
    OFF(sdr_results, listeners_cap);
    OFF(sdr_results, n_listeners);
    OFF(sdr_results, edges);
    OFF(sdr_results, edges_cap);
    OFF(sdr_results, n_edges);
    OFF(sdr_results, runes);
    OFF(sdr_results, rune_frames);
    OFF(sdr_results, runes_cap);
    OFF(sdr_results, n_runes);
    OFF(sdr_results, runes_dropped);
    OFF(sdr_results, edges_dropped);
    printf("abi %d\n", SDR_ABI_VERSION);
    return 0;
}

#define CHECK(call)                                                          \
    do {                                                                     \
        const int rc_ = (call);                                              \
        if (rc_ != SDR_OK) {                                                 \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, sdr_last_error()); \
            return 1;                                                        \
        }                                                                    \
    } while (0)

enum { MAX_CHUNKS = 64, MAX_PEAKS = 4096, MAX_LISTENERS = 64, MAX_EDGES = 65536, MAX_RUNES = 65536 };

static int deliver(sdr_bank *bank)
{
    static sdr_chunk_result chunks[MAX_CHUNKS];
    static sdr_peak peaks[MAX_PEAKS];
    static sdr_listener_result listeners[MAX_LISTENERS];
    static sdr_edge edges[MAX_EDGES];
    static uint32_t runes[MAX_RUNES], rune_frames[MAX_RUNES];
    sdr_results r;
    memset(&r, 0, sizeof r);
    r.struct_size = (int32_t)sizeof r;
    r.chunks = chunks;
    r.chunks_cap = MAX_CHUNKS;
    r.peaks = peaks;
    r.peaks_cap = MAX_PEAKS;
    r.listeners = listeners;
    r.listeners_cap = MAX_LISTENERS;
    r.edges = edges;
    r.edges_cap = MAX_EDGES;
    r.runes = runes;
    r.rune_frames = rune_frames;
    r.runes_cap = MAX_RUNES;
    CHECK(sdr_poll(bank, &r, 1));
    printf("batch %lld first_frame %lld frames %d\n", (long long)r.batch_index, (long long)r.first_frame, (int)r.n_frames);
    for (int c = 0; c < r.n_chunks; c++) {
        printf("chunk frame %lld found %d\n", (long long)chunks[c].frame, (int)chunks[c].peaks_found);
        for (int p = 0; p < chunks[c].n_peaks; p++) {
            const sdr_peak *k = &peaks[chunks[c].first_peak + p];
            uint32_t bits;
            memcpy(&bits, &k->signal_value, 4);
            printf("peak %d %d %d %lld %lld %lld %08x\n", (int)k->from, (int)k->to, (int)k->signal_bin, (long long)k->from_frequency,
                   (long long)k->to_frequency, (long long)k->signal_frequency, (unsigned)bits);
        }
    }
    for (int l = 0; l < r.n_listeners; l++) {
        printf("listener %d edges", (int)listeners[l].listener);
        for (int e = 0; e < listeners[l].n_edges; e++)
            printf(" %u:%u", (unsigned)edges[listeners[l].first_edge + e].frame, (unsigned)edges[listeners[l].first_edge + e].state);
        printf("\nlistener %d runes", (int)listeners[l].listener);
        for (int k = 0; k < listeners[l].n_runes; k++)
            printf(" %u", (unsigned)runes[listeners[l].first_rune + k]);
        printf("\n");
    }
    printf("dropped %llu %llu\n", (unsigned long long)r.runes_dropped, (unsigned long long)r.edges_dropped);
    return 0;
}

static int run(int argc, char **argv)
{
    if (argc < 8) {
        fprintf(stderr, "usage: %s run <iq.f32> <rate> <n> <frames> <edge> <bin> ...\n", argv[0]);
        return 2;
    }
    const int rate = atoi(argv[3]), n = atoi(argv[4]), frames = atoi(argv[5]), edge = atoi(argv[6]);
    const size_t n_floats = (size_t)frames * (size_t)n * 2u;
    float *iq = (float *)malloc(n_floats * sizeof(float));
    FILE *f = fopen(argv[2], "rb");
    if (!iq || !f || fread(iq, sizeof(float), n_floats, f) != n_floats) {
        fprintf(stderr, "cannot read %zu floats from %s\n", n_floats, argv[2]);
        return 2;
    }
    fclose(f);
    sdr_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = (int32_t)sizeof cfg;
    cfg.n_bands = 1;
    cfg.sample_rate = rate;
    cfg.block_size = n;
    cfg.edge_width = edge;
    cfg.peak_threshold = SDR_DEFAULT_PEAK_THRESHOLD;
    cfg.signal_debounce = 1;
    cfg.max_listeners = MAX_LISTENERS;
    cfg.max_batch_frames = frames;
    cfg.max_peaks = 256;
    cfg.find_peaks = 1;
    sdr_bank *bank = NULL;
    CHECK(sdr_create(&cfg, &bank));
    CHECK(sdr_enable_results(bank, 1));
    /* the reference drops a block of the wrong rate or size (rx/receiver.go:319-326): the boundary says so */
    if (sdr_push_iq(bank, 0, rate + 1, iq, (size_t)n * 2u) != SDR_ERR_BAD_RATE || sdr_push_iq(bank, 0, rate, iq, (size_t)n * 2u - 2u) != SDR_ERR_BAD_SIZE) {
        fprintf(stderr, "wrong rate / size not refused\n");
        return 1;
    }
    /* first 100 frames: one cumulation, no listeners yet */
    const int first = frames < 100 ? frames : 100;
    int done = 0;
    CHECK(sdr_push_iq(bank, 0, rate, iq, (size_t)first * (size_t)n * 2u));
    CHECK(sdr_process_staged(bank, &done));
    if (done != first) {
        fprintf(stderr, "processed %d of %d frames\n", done, first);
        return 1;
    }
    if (deliver(bank))
        return 1;
    for (int a = 7; a < argc; a++) {
        int id = -1;
        CHECK(sdr_attach(bank, 0, atoi(argv[a]), &id));
        printf("attached %d -> %d\n", atoi(argv[a]), id);
    }
    if (frames > first) {
        CHECK(sdr_push_iq(bank, 0, rate, iq + (size_t)first * (size_t)n * 2u, (size_t)(frames - first) * (size_t)n * 2u));
        CHECK(sdr_process_staged(bank, &done));
        if (done != frames - first) {
            fprintf(stderr, "processed %d of %d frames\n", done, frames - first);
            return 1;
        }
        if (deliver(bank))
            return 1;
    }
    printf("total %lld\n", (long long)sdr_total_frames(bank));
    CHECK(sdr_destroy(bank));
    free(iq);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc >= 2 && strcmp(argv[1], "layout") == 0)
        return layout();
    if (argc >= 2 && strcmp(argv[1], "run") == 0)
        return run(argc, argv);
    fprintf(stderr, "usage: %s layout | run ...\n", argv[0]);
    return 2;
}
