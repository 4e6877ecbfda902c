// capi_results.hip — bulk delivery: every batch ends with two small kernels (k_results.hip) that copy what the batch
// produced - peaks of each completed cumulation, each listener's keying edges, its newly decoded runes with their frames,
// the bank's drop counters - into the batch's block of pinned host memory, used entries only; sdr_poll hands the oldest
// finished batch to caller-owned buffers without draining the pipeline (the consumer side of rx.Reporter and the
// listeners' io.Writer: rx/rx.go:11-17, rx/receiver.go:123,508-539).  Which batch sits where, and who may take it, is
// host/delivery.h (plain C++, exercised without a GPU by tests/host/test_delivery_model.cpp); this file is its HIP
// backend - events, the layout of a block - and the C ABI in front of it.
#include "bank.h"

using namespace sdrcapi;

namespace sdrcapi {

sdr::ResultsLayout make_results_layout(const sdr_bank *b)
{
    const sdr_config &c = b->cfg;
    sdr::ResultsLayout l{};
    l.max_listeners = c.max_listeners;
    l.max_chunks = b->max_chunks;
    l.max_peaks = c.max_peaks;
    l.edge_cap = b->edge_cap;
    l.text_cap = b->text_cap;
    const size_t B = (size_t)c.n_bands, L = (size_t)c.max_listeners, C = (size_t)b->max_chunks;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t at = off;
        off += (bytes + 63) & ~(size_t)63;
        return at;
    };
    l.off_drops = take(sizeof(sdr::DropCounters));
    l.off_peak_counts = take(B * C * 2 * sizeof(int));
    l.off_peaks = take(B * C * (size_t)c.max_peaks * sizeof(sdr::DevPeak));
    l.off_edge_counts = take(B * L * sizeof(uint32_t));
    l.off_edges = take(B * L * (size_t)b->edge_cap * sizeof(sdr_edge));
    l.off_text_counts = take(B * L * sizeof(uint32_t));
    l.off_text = take(B * L * (size_t)b->text_cap * sizeof(uint32_t));
    l.off_text_frames = take(B * L * (size_t)b->text_cap * sizeof(uint32_t));
    l.bytes = off;
    return l;
}


namespace {
// block (pinned set or parked copy) -> the caller's buffers
int deliver_block(sdr_bank *b, const unsigned char *blk, const host::BatchMeta &m, sdr_results *r)
{
    const sdr_config &c = b->cfg;
    const sdr::ResultsLayout &lay = b->res_layout;
    const int B = c.n_bands, L = c.max_listeners;
    const int *peak_counts = reinterpret_cast<const int *>(blk + lay.off_peak_counts);
    const sdr::DevPeak *peaks = reinterpret_cast<const sdr::DevPeak *>(blk + lay.off_peaks);
    const uint32_t *edge_counts = reinterpret_cast<const uint32_t *>(blk + lay.off_edge_counts);
    const sdr_edge *edges = reinterpret_cast<const sdr_edge *>(blk + lay.off_edges);
    const uint32_t *text_counts = reinterpret_cast<const uint32_t *>(blk + lay.off_text_counts);
    const uint32_t *text = reinterpret_cast<const uint32_t *>(blk + lay.off_text);
    const uint32_t *text_frames = reinterpret_cast<const uint32_t *>(blk + lay.off_text_frames);
    // what is needed
    int64_t need_peaks = 0, need_edges = 0, need_runes = 0;
    int need_listeners = 0;
    for (int band = 0; band < B; band++) {
        for (int ch = 0; ch < m.chunks; ch++)
            need_peaks += peak_counts[2 * ((size_t)band * lay.max_chunks + ch)];
        for (int l = 0; l < m.slots; l++) {
            const size_t idx = (size_t)band * L + l;
            need_edges += edge_counts[idx];
            need_runes += text_counts[idx];
            need_listeners += (edge_counts[idx] || text_counts[idx]) ? 1 : 0;
        }
    }
    const int need_chunks = m.chunks * B;
    const bool fits = need_chunks <= r->chunks_cap && need_peaks <= r->peaks_cap && need_listeners <= r->listeners_cap &&
                      need_edges <= r->edges_cap && need_runes <= r->runes_cap &&
                      (need_chunks == 0 || r->chunks) && (need_peaks == 0 || r->peaks) &&
                      (need_listeners == 0 || r->listeners) && (need_edges == 0 || r->edges) &&
                      (need_runes == 0 || (r->runes && r->rune_frames));
    r->n_chunks = need_chunks;
    r->n_peaks = (int32_t)need_peaks;
    r->n_listeners = need_listeners;
    r->n_edges = (int32_t)need_edges;
    r->n_runes = (int32_t)need_runes;
    r->n_frames = m.frames;
    r->batch_index = m.batch;
    r->first_frame = m.first_frame;
    const sdr::DropCounters *dc = reinterpret_cast<const sdr::DropCounters *>(blk + lay.off_drops);
    r->runes_dropped = dc->runes;
    r->edges_dropped = dc->edges;
    if (!fits)
        return fail(SDR_ERR_BAD_SIZE, "sdr_poll: a result buffer is too small (the n_* fields say what is needed)");
    int ci = 0, pi = 0, li = 0, ei = 0, ri = 0;
    for (int band = 0; band < B; band++) {
        host::FrequencyMapping fm(c.sample_rate, c.block_size, m.center[(size_t)band]);
        for (int ch = 0; ch < m.chunks; ch++) {
            const size_t cidx = (size_t)band * lay.max_chunks + ch;
            const int n = peak_counts[2 * cidx];
            sdr_chunk_result &cr = r->chunks[ci++];
            cr.band = band;
            cr.n_peaks = n;
            cr.frame = m.first_frame + (SDR_CUMULATION_SIZE - m.count0) + (int64_t)ch * SDR_CUMULATION_SIZE - 1;
            cr.first_peak = pi;
            cr.peaks_found = peak_counts[2 * cidx + 1];
            for (int i = 0; i < n; i++) {
                const sdr::DevPeak &p = peaks[cidx * lay.max_peaks + i];
                sdr_peak &o = r->peaks[pi++];
                o.from = p.from;
                o.to = p.to;
                o.signal_bin = p.signal_bin;
                o.signal_value = p.signal_value;
                o.from_frequency = fm.BinToFrequency(p.from, host::BinFrom);
                o.to_frequency = fm.BinToFrequency(p.to, host::BinTo);
                o.signal_frequency = fm.BinToFrequency(p.signal_bin, host::PeakCenterCorrection(p.signal_bin, c.block_size, p.y1, p.y2, p.y3));
            }
        }
    }
    for (int band = 0; band < B; band++)
        for (int l = 0; l < m.slots; l++) {
            const size_t idx = (size_t)band * L + l;
            const int ne = (int)edge_counts[idx], nr = (int)text_counts[idx];
            if (!ne && !nr)
                continue;
            sdr_listener_result &lr = r->listeners[li++];
            lr.band = band;
            lr.listener = l;
            lr.first_edge = ei;
            lr.n_edges = ne;
            lr.first_rune = ri;
            lr.n_runes = nr;
            if (ne)
                memcpy(r->edges + ei, edges + idx * lay.edge_cap, sizeof(sdr_edge) * (size_t)ne);
            if (nr) {
                memcpy(r->runes + ri, text + idx * lay.text_cap, sizeof(uint32_t) * (size_t)nr);
                memcpy(r->rune_frames + ri, text_frames + idx * lay.text_cap, sizeof(uint32_t) * (size_t)nr);
            }
            ei += ne;
            ri += nr;
        }
    return SDR_OK;
}

struct HipResultsBackend final : host::DeliveryBackend {
    sdr_bank *b;
    explicit HipResultsBackend(sdr_bank *bank) : b(bank) {}
    int wait(void *event) override
    {
        HIP_TRY(hipEventSynchronize(static_cast<hipEvent_t>(event)));
        return SDR_OK;
    }
    int query(void *event) override
    {
        const hipError_t q = hipEventQuery(static_cast<hipEvent_t>(event));
        if (q == hipErrorNotReady)
            return SDR_ERR_WOULD_BLOCK;
        HIP_TRY(q);
        return SDR_OK;
    }
    // Only the used entries of the block are copied (a block is 10 MB at config 3; a host that runs four graph replays
    // ahead of its consumer parks six batches per replay).
    std::unique_ptr<unsigned char[]> copy_used(const unsigned char *src, const host::BatchMeta &p) override
    {
    const sdr::ResultsLayout &lay = b->res_layout;
    const sdr_config &c = b->cfg;
    std::unique_ptr<unsigned char[]> block(new unsigned char[lay.bytes]);
    unsigned char *dst = block.get();
    auto copy = [&](size_t off, size_t bytes) { memcpy(dst + off, src + off, bytes); };
    const size_t B = (size_t)c.n_bands, L = (size_t)c.max_listeners, C = (size_t)lay.max_chunks;
    copy(lay.off_drops, sizeof(sdr::DropCounters));
    copy(lay.off_peak_counts, B * C * 2 * sizeof(int));
    copy(lay.off_edge_counts, B * L * sizeof(uint32_t));
    copy(lay.off_text_counts, B * L * sizeof(uint32_t));
    const int *peak_counts = reinterpret_cast<const int *>(src + lay.off_peak_counts);
    const uint32_t *edge_counts = reinterpret_cast<const uint32_t *>(src + lay.off_edge_counts);
    const uint32_t *text_counts = reinterpret_cast<const uint32_t *>(src + lay.off_text_counts);
    for (size_t band = 0; band < B; band++) {
        for (size_t ch = 0; ch < (size_t)p.chunks && ch < C; ch++) {
            const size_t idx = band * C + ch;
            const size_t n = (size_t)std::min(std::max(peak_counts[2 * idx], 0), lay.max_peaks);
            copy(lay.off_peaks + idx * (size_t)lay.max_peaks * sizeof(sdr::DevPeak), n * sizeof(sdr::DevPeak));
        }
        for (size_t l = 0; l < (size_t)p.slots && l < L; l++) {
            const size_t idx = band * L + l;
            const size_t ne = std::min<size_t>(edge_counts[idx], (size_t)lay.edge_cap), nr = std::min<size_t>(text_counts[idx], (size_t)lay.text_cap);
            copy(lay.off_edges + idx * (size_t)lay.edge_cap * sizeof(sdr_edge), ne * sizeof(sdr_edge));
            copy(lay.off_text + idx * (size_t)lay.text_cap * sizeof(uint32_t), nr * sizeof(uint32_t));
            copy(lay.off_text_frames + idx * (size_t)lay.text_cap * sizeof(uint32_t), nr * sizeof(uint32_t));
        }
    }
        return block;
    }
    int deliver(const unsigned char *block, const host::BatchMeta &m, void *out) override { return deliver_block(b, block, m, static_cast<sdr_results *>(out)); }
    int report(int code, const char *msg) override { return sdrcapi::fail(code, msg); }
};
}  // namespace

std::unique_ptr<host::DeliveryBackend> make_results_backend(sdr_bank *b) { return std::unique_ptr<host::DeliveryBackend>(new HipResultsBackend(b)); }

// the set's block of pinned host memory and the two events recorded behind the kernels that fill it
int results_attach_set(sdr_bank *b, int set_idx)
{
    host::ResultSet &rs = b->results->set(set_idx);
    if (!b->res_layout.bytes || (rs.block && rs.ev_listen && rs.ev_peaks))
        return SDR_OK;
    // (the events first, the block last: a set is "attached" only with all three - a failure half way leaves nothing a
    // later call would take for complete)
    if (!rs.ev_listen) {
        hipEvent_t ev = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        rs.ev_listen = ev;
    }
    if (!rs.ev_peaks) {
        hipEvent_t ev = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        rs.ev_peaks = ev;
    }
    if (!rs.block) {
        unsigned char *blk = nullptr;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&blk), b->res_layout.bytes, hipHostMallocDefault));
        memset(blk, 0, b->res_layout.bytes);
        rs.block = blk;
    }
    return SDR_OK;
}

}  // namespace sdrcapi

extern "C" {
#pragma GCC visibility push(default)

int sdr_enable_results(sdr_bank *b, int on)
{
    if (!b)
        return fail(SDR_ERR_BAD_ARG, "null bank");
    int rc = sync_bank(b);
    if (rc)
        return rc;
    if (on) {
        HIP_TRY(hipSetDevice(b->device));
        if (!b->res_layout.bytes)
            b->res_layout = make_results_layout(b);
        b->results->grow((int)b->set.size());
        for (int i = 0; i < (int)b->set.size(); i++)
            if ((rc = results_attach_set(b, i)))
                return rc;
    }
    b->results_on = on != 0;
    b->results->reset(on != 0, b->batch_index);  // (undelivered batches are discarded with the mode)
    return SDR_OK;
}

int sdr_results_pending(sdr_bank *b) { return (b && b->results_on) ? b->results->pending() : 0; }

int sdr_poll(sdr_bank *b, sdr_results *r, int wait)
{
    if (!b || !r)
        return fail(SDR_ERR_BAD_ARG, "null argument");
    if (r->struct_size != (int32_t)sizeof(sdr_results))
        return fail(SDR_ERR_BAD_ARG, "sdr_results.struct_size mismatch (ABI)");
    if (!b->results->on())
        return fail(SDR_ERR_STATE, "bulk delivery is off (sdr_enable_results)");
    HIP_TRY(hipSetDevice(b->device));
    return b->results->poll(r, wait != 0);
}

int sdr_poll_peaks(sdr_bank *b, sdr_results *r, int wait)
{
    if (!b || !r)
        return fail(SDR_ERR_BAD_ARG, "null argument");
    if (r->struct_size != (int32_t)sizeof(sdr_results))
        return fail(SDR_ERR_BAD_ARG, "sdr_results.struct_size mismatch (ABI)");
    if (!b->results_on || !b->listen_pending)
        return fail(SDR_ERR_STATE, "no batch waits for its listen half");
    HIP_TRY(hipSetDevice(b->device));
    return b->results->poll_peaks(b->pend.set, r, wait != 0);
}

int sdr_read_drop_counters(sdr_bank *b, uint64_t *runes_dropped, uint64_t *edges_dropped)
{
    if (!b)
        return fail(SDR_ERR_BAD_ARG, "null bank");
    int rc = sync_bank(b);
    if (rc)
        return rc;
    sdr::DropCounters dc{};
    HIP_TRY(hipMemcpy(&dc, b->drops.p, sizeof dc, hipMemcpyDeviceToHost));
    if (runes_dropped)
        *runes_dropped = dc.runes;
    if (edges_dropped)
        *edges_dropped = dc.edges;
    return SDR_OK;
}

#pragma GCC visibility pop
}  // extern "C"
