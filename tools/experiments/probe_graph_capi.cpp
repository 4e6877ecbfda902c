// Probe: sdr_graph_capture through the C ABI without Python / torch in the process.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../include/sdrainer_hip.h"
int main()
{
    sdr_config c{};
    c.struct_size = sizeof c; c.n_bands = 1; c.sample_rate = 96000; c.block_size = 1024; c.edge_width = 140; c.peak_threshold = 15.f;
    c.signal_debounce = 1; c.max_listeners = 5; c.max_batch_frames = 130; c.max_peaks = 128; c.find_peaks = 1; c.trace = 0; c.device_id = 0;
    sdr_bank *b = nullptr;
    printf("create %d\n", sdr_create(&c, &b));
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    printf("set_stream %d\n", sdr_set_stream(b, s));
    int lid;
    for (int i = 0; i < 5; i++) sdr_attach(b, 0, 200 + 50 * i, &lid);
    printf("enable_results %d\n", sdr_enable_results(b, 1));
    fflush(stdout);
    int rc = sdr_graph_capture(b, 130);
    printf("graph_capture %d %s\n", rc, rc ? sdr_last_error() : "");
    const int K = sdr_graph_batches(b);
    float *iq;
    hipMalloc(&iq, (size_t)K * 130 * 1024 * 8);
    hipMemset(iq, 0, (size_t)K * 130 * 1024 * 8);
    std::vector<const float *> ptrs;
    for (int k = 0; k < K; k++) ptrs.push_back(iq + (size_t)k * 130 * 2048);
    for (int rep = 0; rep < 3; rep++) {
        rc = sdr_graph_launch(b, ptrs.data());
        printf("graph_launch %d %s\n", rc, rc ? sdr_last_error() : "");
    }
    printf("sync %d\n", sdr_sync(b));
    sdr_destroy(b);
    return 0;
}
