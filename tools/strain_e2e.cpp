// Development aid: end-to-end strain mode through the C++ host mirror (rx::Receiver over the C ABI) on device-resident
// IQ - discover -> attach -> decode -> callsign spots - timed in its two regimes: while the listener pool is still
// filling (one listener bound per 100-frame cumulation, rx/receiver.go:409-426: the spectral half of a long segment
// first, the boundary decisions next, the listeners last - rx.h discoverAhead; with SDR_RX_NO_SPECULATION=1 the
// round-2 way, a 100-frame segment resolved on the host before the next) and once it is full (segments of max_batch
// frames, up to four in flight).
// Built as a shared library and driven by tools/strain_e2e.py, which owns the device buffer.
#include <chrono>
#include <cstdio>

#include "../sdrainer_amd/csrc/host/rx.h"

namespace {
struct CountingReporter : rx::Reporter {
    long activated = 0, deactivated = 0, decoded = 0, spotted = 0;
    void ListenerActivated(const std::string &, int64_t) override { activated++; }
    void ListenerDeactivated(const std::string &, int64_t) override { deactivated++; }
    void CallsignDecoded(const std::string &, const std::string &, int64_t, int, int) override { decoded++; }
    void CallsignSpotted(const std::string &, const std::string &, int64_t) override { spotted++; }
};
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

// iq_dev: [frames][2 n] float32 in HBM, replayed until `total_frames` frames have been processed after the pool is full.
// out: [0] frames while hunting, [1] seconds while hunting, [2] frames with the pool full, [3] seconds with the pool
// full, [4] listeners bound, [5] runes decoded, [6] callsigns decoded, [7] callsigns spotted
extern "C" int strain_e2e(const float *iq_dev, int frames, int rate, int n, int pool, int max_batch, long total_frames, double *out)
{
    CountingReporter rep;
    rx::Receiver r("rx", rx::StrainMode, nullptr, pool);
    r.AddReporter(&rep);
    r.SetCenterFrequency(14000000);
    r.SetSelectionPolicy(rx::PeaksTable::StrongestFirst);
    r.SetSilenceTimeout(1e9);
    r.SetAttachmentTimeout(1e9);
    r.SetEdgeWidth(70 * n / 512);
    int rc = r.Start(rate, n, max_batch);
    if (rc != SDR_OK) {
        fprintf(stderr, "Start: %s\n", sdr_last_error());
        return rc;
    }
    // phase 1: hunting.  The buffer is handed over whole (cumulation-aligned: `frames` is a multiple of 100 or the
    // replay starts over at a boundary), the receiver cuts it into segments itself.
    double t0 = now_s();
    long hunted = 0;
    const int whole = frames - frames % 100;
    while (r.Listeners().Available() && hunted < 400L * pool) {
        rc = r.ProcessDevice(iq_dev, whole, false);
        if (rc != SDR_OK)
            return rc;
        hunted += whole;
    }
    rc = r.Flush();
    if (rc != SDR_OK)
        return rc;
    sdr_sync(r.Bank());
    double t1 = now_s();
    fprintf(stderr, "hunting: %.2f ms in all; deciding ahead: %.2f ms waiting for the peaks, %.2f ms deciding and binding, %.2f ms enqueueing the listeners\n",
            (t1 - t0) * 1e3, r.AheadTiming()[0] * 1e3, r.AheadTiming()[1] * 1e3, r.AheadTiming()[2] * 1e3);
    // phase 2: pool full
    long full = 0;
    while (full < total_frames) {
        rc = r.ProcessDevice(iq_dev, frames, false);  // results of the last segments arrive with the next call
        if (rc != SDR_OK)
            return rc;
        full += frames;
    }
    rc = r.Flush();
    if (rc != SDR_OK)
        return rc;
    sdr_sync(r.Bank());
    double t2 = now_s();
    fprintf(stderr, "pool full: %.2f ms in all; resolving segments: %.2f ms in sdr_poll (waiting for the device included), %.2f ms feeding the text processors, %.2f ms replaying their events\n",
            (t2 - t1) * 1e3, r.SegmentTiming()[0] * 1e3, r.SegmentTiming()[1] * 1e3, r.SegmentTiming()[2] * 1e3);
    long runes = 0;
    for (auto &l : r.Listeners().Listeners())
        runes += (long)l->Text().size();
    out[0] = (double)hunted;
    out[1] = t1 - t0;
    out[2] = (double)full;
    out[3] = t2 - t1;
    out[4] = (double)r.Listeners().Listeners().size();
    out[5] = (double)runes;
    out[6] = (double)rep.decoded;
    out[7] = (double)rep.spotted;
    return SDR_OK;
}
