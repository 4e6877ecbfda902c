// rx.h — C++ host mirror of sdrainer's rx package for the hot path, written against the C ABI only
// (include/sdrainer_hip.h).  Same names, argument meaning and error behaviour as the reference so the
// tests read like rx/peaks_test.go, rx/listener_test.go:
//
//   rx::Clock / ManualClock      rx/receiver.go:29-55
//   rx::Reporter                 rx/rx.go:11-17
//   rx::PeaksTable               rx/peaks.go (whole file)
//   rx::IDPool, ListenerPool     rx/listener.go:150-270
//   rx::Listener                 rx/listener.go:19-148 (the demodulator lives on the GPU)
//   rx::Receiver                 rx/receiver.go:64-500 (run loop = Process())
//
// Pure bookkeeping stays on the host exactly as in the reference: which peaks are known, which
// listener is bound to which peak, time-outs.  Everything per frame is on the device.
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <memory>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/sdrainer_hip.h"
#include "frequency_mapping.h"
#include "text_processor.h"

namespace rx {

// ---------------------------------------------------------------------------------------------
// Clock (rx/receiver.go:29-55).  Time is in seconds.
// ---------------------------------------------------------------------------------------------
struct Clock {
    virtual ~Clock() = default;
    virtual double Now() const = 0;
};
struct ManualClock : Clock {  // rx/receiver.go:41-55
    double now = 0;
    double Now() const override { return now; }
    void Set(double t) { now = t; }
    void Add(double d) { now += d; }
};

// rx/receiver.go:15-27 and rx/listener.go:14-17, rx/peaks.go:10-12
constexpr int kCumulationSize = SDR_CUMULATION_SIZE;
constexpr int kPeakPadding = 0;
constexpr double kDefaultPeakTimeout = 120.0;
constexpr double kDefaultSilenceTimeout = 20.0;
constexpr double kDefaultAttachmentTimeout = 120.0;

using Peak = sdr_peak;  // dsp.Peak[float32,int] (dsp/fft.go:179-188)

// rx/rx.go:11-17
struct Reporter {
    virtual ~Reporter() = default;
    virtual void ListenerActivated(const std::string &listener, int64_t frequency) = 0;
    virtual void ListenerDeactivated(const std::string &listener, int64_t frequency) = 0;
    virtual void CallsignDecoded(const std::string &, const std::string &, int64_t, int, int) {}
    virtual void CallsignSpotted(const std::string &, const std::string &, int64_t) {}
    virtual void SpotTimeout(const std::string &, const std::string &, int64_t) {}
};

// ---------------------------------------------------------------------------------------------
// PeaksTable (rx/peaks.go)
// ---------------------------------------------------------------------------------------------
enum PeakState { peakNone = 0, peakNew, peakActive, peakInactive };  // rx/peaks.go:23-28

class PeaksTable {
public:
    struct Entry {  // rx/peaks.go:14-19
        Peak peak;
        PeakState state;
        double since;
    };
    // FindNext picks a random new peak in the reference (unseeded math/rand, rx/peaks.go:185).  The
    // probe sequence is injectable so runs are reproducible; nullptr = deterministic linear scan only.
    using RandFn = std::function<int(int)>;

    PeaksTable(int size, const Clock *clock) : bins_((size_t)size, nullptr), clock_(clock) {}

    void SetRand(RandFn f) { rand_ = std::move(f); }
    // Selection among the new peaks: the reference's random probe followed by a linear scan (default; the
    // scan alone when no RandFn is set), or the strongest new peak first (ties: lowest bin) - a policy
    // whose runs are reproducible at hundreds of listeners without a seed.
    enum Policy { ReferenceOrder, StrongestFirst };
    void SetPolicy(Policy p) { policy_ = p; }
    void SetPeakTimeout(double seconds) { peakTimeout_ = seconds; }

    void ForcePut(const Peak &p) { put(p, true); }  // :46-71
    void Put(const Peak &p) { put(p, false); }      // :73-103

    const Peak *Get(int bin) const  // :107-117
    {
        if (bin < 0 || bin >= (int)bins_.size() || !bins_[bin])
            return nullptr;
        return &bins_[bin]->peak;
    }
    const Entry *GetEntry(int bin) const { return (bin < 0 || bin >= (int)bins_.size()) ? nullptr : bins_[bin].get(); }

    void Cleanup()  // :127-147
    {
        const double now = clock_->Now();
        int i = 0;
        while (i < (int)bins_.size()) {
            const Entry *p = bins_[i].get();
            i++;
            if (!p || p->state == peakActive || now - p->since < peakTimeout_)
                continue;
            const int from = p->peak.from, to = p->peak.to;  // (clear drops the entry)
            clear(from, to);
            i = to + 1;
        }
    }
    void Reset() { std::fill(bins_.begin(), bins_.end(), nullptr); }  // :149-151

    void Activate(const Peak &p)  // :153-159 (the reference nil-derefs on an unknown peak, App. C4)
    {
        Entry *e = getInternal(p);
        if (e && (e->state == peakNew || e->state == peakInactive))
            e->state = peakActive;
    }
    void Deactivate(const Peak &p)  // :174-181
    {
        Entry *e = getInternal(p);
        if (e && e->state == peakActive)
            e->state = peakInactive;
    }
    const Peak *FindNext()  // :183-207
    {
        const int n = (int)bins_.size();
        if (policy_ == StrongestFirst) {
            const Entry *best = nullptr;
            for (auto &p : bins_)
                if (p && p->state == peakNew && (!best || p->peak.signal_value > best->peak.signal_value))
                    best = p.get();
            return best ? &best->peak : nullptr;
        }
        if (rand_)
            for (int k = 0; k < n / 2; k++) {
                const int i = rand_(n);
                if (bins_[i] && bins_[i]->state == peakNew)
                    return &bins_[i]->peak;
            }
        for (auto &p : bins_)
            if (p && p->state == peakNew)
                return &p->peak;
        return nullptr;
    }
    int Size() const { return (int)bins_.size(); }

private:
    void put(const Peak &p, bool force)
    {
        const int n = (int)bins_.size();
        int clearFrom = -1, clearTo = -1;
        for (int i = std::max(0, (int)p.from); i <= std::min((int)p.to, n - 1); i++) {
            const std::shared_ptr<Entry> &e = bins_[i];
            if (!e)
                continue;
            if (!force && (e->state == peakActive || e->state == peakInactive))
                return;
            if (clearFrom == -1)
                clearFrom = e->peak.from;
            clearTo = e->peak.to;
        }
        if (clearFrom > -1 && clearTo > -1)
            clear(clearFrom, clearTo);
        auto e = std::make_shared<Entry>(Entry{p, peakNew, clock_->Now()});
        for (int i = std::max(0, (int)p.from); i <= std::min((int)p.to, n - 1); i++)  // :105-109
            bins_[i] = e;
    }
    void clear(int from, int to)  // :111-115
    {
        for (int i = std::max(0, from); i <= std::min(to, (int)bins_.size() - 1); i++)
            bins_[i] = nullptr;
    }
    Entry *getInternal(const Peak &p)  // :161-172
    {
        if (p.from < 0 || p.from >= (int)bins_.size() || !bins_[p.from] || bins_[p.from]->peak.to != p.to)
            return nullptr;
        return bins_[p.from].get();
    }

    std::vector<std::shared_ptr<Entry>> bins_;
    const Clock *clock_;
    double peakTimeout_ = kDefaultPeakTimeout;
    RandFn rand_;
    Policy policy_ = ReferenceOrder;
};

// ---------------------------------------------------------------------------------------------
// IDPool / Listener / ListenerPool (rx/listener.go)
// ---------------------------------------------------------------------------------------------
class IDPool {  // :150-178
public:
    IDPool(int size, const std::string &prefix)
    {
        for (int i = 0; i < size; i++)
            ids_.push_back(prefix + std::to_string(size - i));
    }
    void Push(const std::string &id) { ids_.push_back(id); }
    bool Pop(std::string *out)
    {
        if (ids_.empty())
            return false;
        *out = ids_.back();
        ids_.pop_back();
        return true;
    }
    size_t Len() const { return ids_.size(); }

private:
    std::vector<std::string> ids_;
};

class Listener : private CallsignReporter {  // :19-148
public:
    Listener(std::string id, const Clock *clock, Reporter *reporter)
        : id_(std::move(id)), clock_(clock), reporter_(reporter),
          textProcessor_([this] { return timeOverride_ ? overrideNow_ : (clock_ ? clock_->Now() : 0.0); }, this)
    {
    }
    // While the receiver feeds a segment's runes (possibly several listeners in parallel) each Write happens at
    // the time of the rune's frame, and the reporter calls it triggers are kept until FlushEvents().
    void BeginFeed() { deferEvents_ = true; }
    void SetFeedTime(double t)
    {
        timeOverride_ = true;
        overrideNow_ = t;
    }
    void EndFeed() { timeOverride_ = false; }
    void FlushEvents()
    {
        deferEvents_ = false;
        for (const Deferred &e : deferred_) {
            if (e.kind == 0)
                reporter_->CallsignDecoded(id_, e.callsign, e.frequency, e.count, e.weight);
            else if (e.kind == 1)
                reporter_->CallsignSpotted(id_, e.callsign, e.frequency);
            else
                reporter_->SpotTimeout(id_, e.callsign, e.frequency);
        }
        deferred_.clear();
    }
    Listener(const Listener &) = delete;
    Listener &operator=(const Listener &) = delete;
    const std::string &ID() const { return id_; }
    void SetSilenceTimeout(double s) { silenceTimeout_ = s; }
    void SetAttachmentTimeout(double s) { attachmentTimeout_ = s; }

    void Attach(const Peak &peak, int device_id)  // :84-94
    {
        peak_ = peak;
        attached_ = true;
        device_id_ = device_id;
        lastAttach_ = clock_->Now();
        textProcessor_.Restart();  // :89
        text_.clear();
        if (reporter_)
            reporter_->ListenerActivated(id_, peak_.signal_frequency);
    }
    bool Attached() const { return attached_; }  // :96-98
    void Detach()                                // :99-108
    {
        const int64_t f = peak_.signal_frequency;
        attached_ = false;
        if (reporter_)
            reporter_->ListenerDeactivated(id_, f);
    }
    const Peak &GetPeak() const { return peak_; }
    int SignalBin() const { return attached_ ? peak_.signal_bin : 0; }  // :119-124
    int DeviceID() const { return device_id_; }
    bool TimeoutExceeded() const  // :126-136
    {
        const double now = clock_->Now();
        return (now - lastAttach_ > attachmentTimeout_) || (now - textProcessor_.LastWrite() > silenceTimeout_);
    }
    void CheckWriteTimeout() { textProcessor_.CheckWriteTimeout(); }  // :138-140
    // The io.Writer the GPU decoder's runes arrive at (TextProcessor.Write, rx/text_processor.go:202-216);
    // `text_` plays the part of the reference's `out` writer.
    void Write(const std::string &utf8)
    {
        // the decoder writes one rune per call (cw/decode.go:352); keep that granularity, the window's
        // one-FindNext-per-Write cadence depends on it
        for (size_t i = 0; i < utf8.size();) {
            size_t len = 1;
            const unsigned char c = (unsigned char)utf8[i];
            if (c >= 0xF0)
                len = 4;
            else if (c >= 0xE0)
                len = 3;
            else if (c >= 0xC0)
                len = 2;
            len = std::min(len, utf8.size() - i);
            textProcessor_.Write(utf8.substr(i, len));
            i += len;
        }
        text_ += utf8;
    }
    void WriteRune(uint32_t r)  // one Decoder write (cw/decode.go:352: one rune per call)
    {
        char b[4];
        size_t n = 0;
        if (r < 0x80) {
            b[n++] = (char)r;
        } else if (r < 0x800) {
            b[n++] = (char)(0xC0 | (r >> 6));
            b[n++] = (char)(0x80 | (r & 0x3F));
        } else if (r < 0x10000) {
            b[n++] = (char)(0xE0 | (r >> 12));
            b[n++] = (char)(0x80 | ((r >> 6) & 0x3F));
            b[n++] = (char)(0x80 | (r & 0x3F));
        } else {
            b[n++] = (char)(0xF0 | (r >> 18));
            b[n++] = (char)(0x80 | ((r >> 12) & 0x3F));
            b[n++] = (char)(0x80 | ((r >> 6) & 0x3F));
            b[n++] = (char)(0x80 | (r & 0x3F));
        }
        const std::string s(b, n);
        textProcessor_.Write(s);
        text_ += s;
    }
    const std::string &Text() const { return text_; }
    TextProcessor &Processor() { return textProcessor_; }
    double LastAttach() const { return lastAttach_; }
    double LastWrite() const { return textProcessor_.LastWrite(); }
    double SilenceTimeout() const { return silenceTimeout_; }
    double AttachmentTimeout() const { return attachmentTimeout_; }

private:
    // :70-83 — the text processor's callbacks, forwarded with the listener's id and signal frequency
    void CallsignDecoded(const std::string &callsign, int count, int weight) override
    {
        if (!reporter_)
            return;
        const int64_t f = peak_.signal_frequency;
        if (deferEvents_)
            deferred_.push_back(Deferred{0, count, weight, f, callsign});
        else
            reporter_->CallsignDecoded(id_, callsign, f, count, weight);
    }
    void CallsignSpotted(const std::string &callsign) override
    {
        if (!reporter_)
            return;
        const int64_t f = peak_.signal_frequency;
        if (deferEvents_)
            deferred_.push_back(Deferred{1, 0, 0, f, callsign});
        else
            reporter_->CallsignSpotted(id_, callsign, f);
    }
    void SpotTimeout(const std::string &callsign) override
    {
        if (!reporter_)
            return;
        const int64_t f = peak_.signal_frequency;
        if (deferEvents_)
            deferred_.push_back(Deferred{2, 0, 0, f, callsign});
        else
            reporter_->SpotTimeout(id_, callsign, f);
    }

    std::string id_;
    const Clock *clock_;
    Reporter *reporter_;
    // (in front of the text processor: its constructor asks for the time, and the clock lambda reads these)
    bool timeOverride_ = false, deferEvents_ = false;
    double overrideNow_ = 0;
    TextProcessor textProcessor_;
    Peak peak_{};
    bool attached_ = false;
    int device_id_ = -1;
    double lastAttach_ = 0;
    double silenceTimeout_ = kDefaultSilenceTimeout, attachmentTimeout_ = kDefaultAttachmentTimeout;
    std::string text_;
    struct Deferred {  // a reporter call kept until FlushEvents (a record, not a closure: there are hundreds per segment)
        int kind, count, weight;
        int64_t frequency;
        std::string callsign;
    };
    std::vector<Deferred> deferred_;
};

class ListenerPool {  // :180-270
public:
    using Factory = std::function<std::shared_ptr<Listener>(const std::string &id)>;
    ListenerPool(int size, const std::string &prefix, Factory f) : size_(size), ids_(size, prefix), factory_(std::move(f)) {}
    int Size() const { return size_; }
    bool Available() const { return (int)listeners_.size() < size_; }
    void Reset()
    {
        for (auto &l : listeners_) {
            l->Detach();
            ids_.Push(l->ID());
        }
        listeners_.clear();
    }
    std::shared_ptr<Listener> BindNext()  // :214-229
    {
        if ((int)listeners_.size() == size_)
            return nullptr;
        std::string id;
        if (!ids_.Pop(&id))
            return nullptr;
        auto l = factory_(id);
        listeners_.push_back(l);
        return l;
    }
    void Release(const std::shared_ptr<Listener> &l)  // :237-248: swap-remove
    {
        int index = -1;
        for (size_t i = 0; i < listeners_.size(); i++)
            if (listeners_[i]->ID() == l->ID()) {
                index = (int)i;
                break;
            }
        if (index == -1)
            return;
        ids_.Push(l->ID());
        if (listeners_.size() > 1)
            listeners_[index] = listeners_.back();
        listeners_.pop_back();
    }
    const std::vector<std::shared_ptr<Listener>> &Listeners() const { return listeners_; }
    std::shared_ptr<Listener> First() const { return listeners_.empty() ? nullptr : listeners_[0]; }

private:
    int size_;
    std::vector<std::shared_ptr<Listener>> listeners_;
    IDPool ids_;
    Factory factory_;
};

// A few long-lived worker threads for the per-listener text processing (the reference gives every
// TextProcessor a goroutine of its own, rx/text_processor.go:161-171).  Run(n, fn) calls fn(i) for i in [0, n)
// on the workers and the caller, and returns when all are done.
class Workers {
public:
    explicit Workers(int n_threads)
    {
        for (int i = 0; i < n_threads; i++)
            threads_.emplace_back([this] { loop(); });
    }
    ~Workers()
    {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : threads_)
            t.join();
    }
    // fn(0) .. fn(n - 1), each once, on the pool's threads and the caller's; returns when all are done.  The items are
    // short (a listener's runes of one segment: a microsecond or two), so they are handed out a few at a time from an
    // atomic counter - a mutex round trip per item cost more than the items.
    void Run(size_t n, const std::function<void(size_t)> &fn)
    {
        if (n == 0)
            return;
        {
            std::lock_guard<std::mutex> g(m_);
            fn_ = &fn;
            n_ = n;
            next_.store(0, std::memory_order_relaxed);
            done_.store(0, std::memory_order_relaxed);
            generation_++;
        }
        cv_.notify_all();
        work();
        // (the others finish within microseconds of the caller; a thread that joined this run has left it again
        // before the function and the counters may be reused)
        for (;;) {
            if (done_.load(std::memory_order_acquire) >= n) {
                std::lock_guard<std::mutex> g(m_);
                if (active_ == 0) {
                    fn_ = nullptr;
                    return;
                }
            }
            std::this_thread::yield();
        }
    }

private:
    static constexpr size_t kChunk = 4;
    void work()
    {
        const std::function<void(size_t)> *fn;
        size_t n;
        {
            std::lock_guard<std::mutex> g(m_);
            if (!fn_)
                return;  // (woken after the run it was woken for had finished)
            fn = fn_;
            n = n_;
            active_++;
        }
        for (;;) {
            const size_t i0 = next_.fetch_add(kChunk, std::memory_order_relaxed);
            if (i0 >= n)
                break;
            const size_t i1 = std::min(n, i0 + kChunk);
            for (size_t i = i0; i < i1; i++)
                (*fn)(i);
            done_.fetch_add(i1 - i0, std::memory_order_release);
        }
        std::lock_guard<std::mutex> g(m_);
        active_--;
    }
    void loop()
    {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return stop_ || generation_ != seen; });
                if (stop_)
                    return;
                seen = generation_;
            }
            work();
        }
    }
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable cv_;
    const std::function<void(size_t)> *fn_ = nullptr;
    size_t n_ = 0;
    std::atomic<size_t> next_{0}, done_{0};
    int active_ = 0;
    uint64_t generation_ = 0;
    bool stop_ = false;
};

// ---------------------------------------------------------------------------------------------
// Receiver (rx/receiver.go)
// ---------------------------------------------------------------------------------------------
enum ReceiverMode { DecodeMode, StrainMode };  // :57-62

class Receiver {
public:
    // `clock` drives peak / listener time-outs.  The reference uses wall time; a stream clock (frames
    // processed x blockSize / sampleRate) makes offline runs reproducible: pass nullptr to get one.
    Receiver(std::string id, ReceiverMode mode, Clock *clock = nullptr, int listenerPoolSize = SDR_DEFAULT_LISTENER_POOL_SIZE)
        : id_(std::move(id)), mode_(mode), clock_(clock ? clock : &streamClock_),
          listeners_(mode == DecodeMode ? 1 : listenerPoolSize, id_, [this](const std::string &lid) { return newListener(lid); })
    {
    }
    ~Receiver() { Stop(); }

    void AddReporter(Reporter *r) { reporters_.push_back(r); }
    // setters before Start are stored, after Start they are forwarded between batches (:166-172)
    void SetPeakThreshold(float t)
    {
        peakThreshold_ = t;
        if (bank_)
            sdr_set_peak_threshold(bank_, 0, t);
    }
    void SetEdgeWidth(int e)
    {
        edgeWidth_ = e;
        if (bank_)
            sdr_set_edge_width(bank_, e);
    }
    void SetSilenceTimeout(double s)
    {
        silenceTimeout_ = s;
        for (auto &l : listeners_.Listeners())
            l->SetSilenceTimeout(s);
    }
    void SetAttachmentTimeout(double s)
    {
        attachmentTimeout_ = s;
        for (auto &l : listeners_.Listeners())
            l->SetAttachmentTimeout(s);
    }
    void SetSignalDebounce(int d)  // :238-244: applies to the CURRENT listeners only, as in the reference
    {
        if (bank_)
            sdr_set_signal_debounce(bank_, 0, d);
    }
    void SetCenterFrequency(int64_t f)
    {
        centerFrequency_ = f;
        if (bank_)
            sdr_set_center_frequency(bank_, 0, f);
    }
    int64_t CenterFrequency() const { return centerFrequency_; }
    void SetSelectionPolicy(PeaksTable::Policy p)
    {
        policy_ = p;
        if (peaks_)
            peaks_->SetPolicy(p);
    }
    void SetFindNextRand(PeaksTable::RandFn f)
    {
        rand_ = std::move(f);
        if (peaks_)
            peaks_->SetRand(rand_);
    }

    int Start(int sampleRate, int blockSize, int maxBatchFrames = 256, int deviceId = 0)  // :130-146
    {
        if (bank_)
            return SDR_OK;
        sdr_config cfg{};
        cfg.struct_size = sizeof cfg;
        cfg.n_bands = 1;
        cfg.sample_rate = sampleRate;
        cfg.block_size = blockSize;
        cfg.edge_width = edgeWidth_;
        cfg.peak_threshold = peakThreshold_;
        cfg.signal_debounce = 1;  // NewSpectralDemodulator: defaultSignalDebounce (cw/spectral.go:14,27)
        cfg.max_listeners = listeners_.Size();
        cfg.max_batch_frames = maxBatchFrames;
        cfg.max_peaks = blockSize / 2;
        cfg.find_peaks = mode_ == StrainMode;
        cfg.device_id = deviceId;
        int rc = sdr_create(&cfg, &bank_);
        if (rc != SDR_OK)
            return rc;
        sampleRate_ = sampleRate;
        blockSize_ = blockSize;
        maxBatchFrames_ = maxBatchFrames;
        sdr_set_center_frequency(bank_, 0, centerFrequency_);
        // results arrive in bulk, one sdr_poll per processed segment (no per-listener reads, no pipeline drain)
        rc = sdr_enable_results(bank_, 1);
        if (rc != SDR_OK)
            return rc;
        const size_t chunks = (size_t)maxBatchFrames / kCumulationSize + 2, pool = (size_t)std::max(listeners_.Size(), 1);
        resChunks_.resize(chunks);
        resPeaks_.resize(chunks * (size_t)cfg.max_peaks);
        resListeners_.resize(pool);
        resEdges_.resize(pool * (size_t)std::min(maxBatchFrames, 8192));
        resRunes_.resize(pool * 2048);
        resRuneFrames_.resize(pool * 2048);
        peaks_.reset(new PeaksTable(blockSize, clock_));
        peaks_->SetRand(rand_);
        peaks_->SetPolicy(policy_);
        return SDR_OK;
    }
    void Stop()  // :148-164
    {
        if (!bank_)
            return;
        listeners_.Reset();
        sdr_destroy(bank_);
        bank_ = nullptr;
    }

    // :315-334 — never throws; wrong rate / size / full queue are logged and dropped
    int IQData(int sampleRate, const float *data, size_t n_floats)
    {
        if (!bank_)
            return SDR_OK;
        const int rc = sdr_push_iq(bank_, 0, sampleRate, data, n_floats);
        if (rc == SDR_ERR_BAD_RATE)
            fprintf(stderr, "wrong incoming sample rate on receiver %s: %d instead of %d!\n", id_.c_str(), sampleRate, sampleRate_);
        else if (rc == SDR_ERR_BAD_SIZE)
            fprintf(stderr, "wrong incoming block size on receiver %s: %zu instead of %d\n", id_.c_str(), n_floats, blockSize_);
        else if (rc == SDR_ERR_WOULD_DROP)
            fprintf(stderr, "IQ data skipped on receiver %s\n", id_.c_str());
        return rc;
    }

    // :272-313 SetVFOOffset, DecodeMode branch: force a peak at the VFO frequency and listen to it
    int SetVFOOffset(int64_t offset)
    {
        vfoOffset_ = offset;
        if (!bank_ || mode_ != DecodeMode)
            return SDR_OK;
        if (!listeners_.Available()) {
            for (auto &l : listeners_.Listeners())
                if (l->Attached())
                    sdr_detach(bank_, 0, l->DeviceID());
            listeners_.Reset();
        }
        auto listener = listeners_.BindNext();
        if (!listener)
            return SDR_ERR_NO_SLOT;
        Peak peak = newPeakCenteredOnFrequency(vfoOffset_ + centerFrequency_);
        peak.signal_value = 80;
        peaks_->ForcePut(peak);
        peaks_->Activate(peak);
        return attach(listener, peak);
    }

    // The frame case of run() (:353-463) for everything staged so far (IQData), or for `n_frames` frames already
    // in device memory ([frame][2 * blockSize] float32, 16-byte aligned).  The stream is processed in segments;
    // where a segment ends is what keeps the per-frame semantics of the reference (:388-426) although the
    // device works on many frames per call:
    //  * strain mode with a free listener: at every cumulation boundary, where the reference binds a new
    //    listener that must start listening with the very next frame (:409-426);
    //  * any listener: at the earliest frame its attachment or silence time-out can fire (:396-400).  With the
    //    stream clock (frame f happens at (f + 1) * blockSize / sampleRate) that frame is known: the attachment
    //    expiry exactly, the silence expiry as "no rune from now on" - runes only postpone it, and every rune
    //    comes back stamped with the frame of its Write (text_processor.go:208-209), so the check after the
    //    segment is the reference's check at exactly that frame.  With a caller-supplied clock the time of a
    //    frame is not defined; time-outs are then evaluated once per segment, at the clock's value.
    int Process()
    {
        if (!bank_)
            return SDR_OK;
        for (;;) {
            const int staged = sdr_staged_frames(bank_, 0);
            if (staged <= 0)
                return resolvePending();
            const int limit = segmentLimit(staged);
            int n = 0;
            int rc = sdr_defer_listen(bank_, speculative_ ? 1 : 0);
            if (rc == SDR_OK)
                rc = sdr_process_staged_limit(bank_, limit, &n);
            if (rc != SDR_OK)
                return rc;
            if (n == 0)
                return resolvePending();
            rc = segmentLaunched(n);
            if (rc != SDR_OK)
                return rc;
        }
    }
    // `drain` = false leaves the last segments in flight (their results arrive with a later call or Flush()): a
    // caller that streams buffer after buffer keeps the device pipeline full across calls.
    int Flush() { return bank_ ? resolvePending() : SDR_OK; }
    int ProcessDevice(const float *iq_dev, int n_frames, bool drain = true)
    {
        if (!bank_)
            return SDR_OK;
        int done = 0;
        while (done < n_frames) {
            const int n = segmentLimit(n_frames - done);
            int rc = sdr_defer_listen(bank_, speculative_ ? 1 : 0);
            if (rc == SDR_OK)
                rc = sdr_process_device(bank_, iq_dev + (size_t)done * 2 * (size_t)blockSize_, n);
            if (rc != SDR_OK)
                return rc;
            rc = segmentLaunched(n);
            if (rc != SDR_OK)
                return rc;
            done += n;
        }
        return drain ? resolvePending() : SDR_OK;
    }

    PeaksTable &Peaks() { return *peaks_; }
    ListenerPool &Listeners() { return listeners_; }
    sdr_bank *Bank() { return bank_; }
    // frames the receiver has been through; inside a Reporter callback: up to and including the frame the event belongs
    // to (the device may be further ahead - segments in flight, or a segment whose boundaries are being decided)
    int64_t FramesProcessed() const { return reportFrames_ >= 0 ? reportFrames_ : framesProcessed_; }
    // seconds spent in discoverAhead so far: waiting for the peaks, deciding + binding, enqueueing the listen half
    const double *AheadTiming() const { return aheadTiming_; }
    // seconds spent resolving segments so far: in sdr_poll (waiting for the device included), feeding the text
    // processors, replaying their reporter events
    const double *SegmentTiming() const { return segmentTiming_; }
    const std::vector<Peak> &LastPeaks() const { return lastPeaks_; }

    // :474-500 (peakPadding = 0: a found run is re-centred to its strongest bin)
    Peak newPeakCenteredOnBin(int centerBin) const
    {
        host::FrequencyMapping fm(sampleRate_, blockSize_, centerFrequency_);
        Peak p{};
        p.from = std::max(0, centerBin - kPeakPadding);
        p.to = std::min(centerBin + kPeakPadding, blockSize_ - 1);
        p.from_frequency = fm.BinToFrequency(p.from, host::BinFrom);
        p.to_frequency = fm.BinToFrequency(p.to, host::BinTo);
        p.signal_frequency = p.from_frequency + (p.to_frequency - p.from_frequency) / 2;  // Peak.CenterFrequency
        return p;
    }
    Peak newPeakCenteredOnSignal(const Peak &peak) const
    {
        Peak r = newPeakCenteredOnBin(peak.signal_bin);
        r.signal_frequency = peak.signal_frequency;
        r.signal_value = peak.signal_value;
        r.signal_bin = peak.signal_bin;
        return r;
    }
    Peak newPeakCenteredOnFrequency(int64_t frequency) const
    {
        host::FrequencyMapping fm(sampleRate_, blockSize_, centerFrequency_);
        const int bin = fm.FrequencyToBin(frequency);
        Peak r = newPeakCenteredOnBin(bin);
        r.signal_bin = bin;
        r.signal_frequency = frequency;
        return r;
    }

private:
    struct Fanout : Reporter {
        Receiver *r;
        void ListenerActivated(const std::string &l, int64_t f) override
        {
            for (auto *rep : r->reporters_)
                rep->ListenerActivated(l, f);
        }
        void ListenerDeactivated(const std::string &l, int64_t f) override
        {
            for (auto *rep : r->reporters_)
                rep->ListenerDeactivated(l, f);
        }
        void CallsignDecoded(const std::string &l, const std::string &c, int64_t f, int count, int weight) override
        {
            for (auto *rep : r->reporters_)
                rep->CallsignDecoded(l, c, f, count, weight);
        }
        void CallsignSpotted(const std::string &l, const std::string &c, int64_t f) override
        {
            for (auto *rep : r->reporters_)
                rep->CallsignSpotted(l, c, f);
        }
        void SpotTimeout(const std::string &l, const std::string &c, int64_t f) override
        {
            for (auto *rep : r->reporters_)
                rep->SpotTimeout(l, c, f);
        }
    };
    std::shared_ptr<Listener> newListener(const std::string &lid)  // :120-126
    {
        fanout_.r = this;
        auto l = std::make_shared<Listener>(lid, clock_, &fanout_);
        l->SetAttachmentTimeout(attachmentTimeout_);
        l->SetSilenceTimeout(silenceTimeout_);
        return l;
    }
    // first_frame >= 0: the frame the listener hears first, inside (or right behind) a segment whose spectra exist
    int attach(const std::shared_ptr<Listener> &listener, const Peak &peak, int64_t first_frame = -1)
    {
        int dev = -1;
        const int rc = first_frame >= 0 && sdr_listen_pending(bank_) ? sdr_attach_at(bank_, 0, peak.signal_bin, first_frame, &dev)
                                                                      : sdr_attach(bank_, 0, peak.signal_bin, &dev);
        if (rc != SDR_OK)
            return rc;
        listener->Attach(peak, dev);
        return SDR_OK;
    }
    double frameTime(int64_t f) const { return (double)(f + 1) * (double)blockSize_ / (double)sampleRate_; }
    bool firesAt(const Listener &l, int64_t f) const  // Listener.TimeoutExceeded with the clock at frame f
    {
        const double now = frameTime(f);
        return (now - l.LastAttach() > l.AttachmentTimeout()) || (now - l.LastWrite() > l.SilenceTimeout());
    }
    // first frame >= from at which the listener's time-out fires if it writes nothing more
    int64_t earliestExpiry(const Listener &l, int64_t from) const
    {
        const double T = (double)blockSize_ / (double)sampleRate_;
        const double due = std::min(l.LastAttach() + l.AttachmentTimeout(), l.LastWrite() + l.SilenceTimeout());
        if (!(due < 1e18))
            return INT64_MAX;
        int64_t f = (int64_t)std::floor(due / T);  // (f + 1) * T > due  <=>  f + 1 > due / T
        f = std::max(f - 2, from);                 // settle rounding with the predicate itself
        while (!firesAt(l, f))
            f++;
        return f;
    }
    // how many of `available` frames the next device call may take
    int segmentLimit(int available)
    {
        int limit = std::min(available, maxBatchFrames_);
        segExpiryAtEnd_ = false;
        speculative_ = false;
        if (mode_ != StrainMode)
            return limit;
        const int until_boundary = kCumulationSize - (int)(framesProcessed_ % kCumulationSize);
        const bool hunting = listeners_.Available();
        // Hunting over a long segment (stream clock only - with a caller's clock the time of a frame inside a segment
        // is not defined): the device runs the spectral half of the whole segment, the host then makes the decisions
        // of every cumulation boundary in it, in order (discoverAhead), and only then the listeners run.  A listener
        // bound at the segment's first boundary must not be able to time out inside the segment: that bounds its length.
        speculative_ = hunting && clock_ == &streamClock_ && !std::getenv("SDR_RX_NO_SPECULATION");
        if (hunting && !speculative_)
            limit = std::min(limit, until_boundary);
        if (speculative_) {
            const double T = (double)blockSize_ / (double)sampleRate_;
            const double shortest = std::min(attachmentTimeout_, silenceTimeout_);
            if (shortest < 1e17) {
                // bound at frame b (clock frameTime(b)), it fires at the first f with (f - b) * T > shortest
                const int64_t quiet = (int64_t)std::floor(shortest / T) - 1;  // frames after b that are certainly safe
                limit = (int)std::max<int64_t>(std::min<int64_t>(limit, until_boundary + std::max<int64_t>(quiet, 0)), 1);
            }
        }
        if (clock_ == &streamClock_) {
            int64_t first = INT64_MAX;
            for (auto &l : listeners_.Listeners())
                if (l->Attached())
                    first = std::min(first, earliestExpiry(*l, framesProcessed_));
            if (first != INT64_MAX && first - framesProcessed_ + 1 <= (int64_t)limit) {
                limit = (int)(first - framesProcessed_ + 1);
                segExpiryAtEnd_ = true;
            }
        }
        // FindPeaks is only needed where a listener can be bound: at a boundary with a free slot (:410) - free
        // now, or freed by a time-out on the segment's last frame
        const bool ends_on_boundary = limit >= until_boundary;
        sdr_set_find_peaks(bank_, (hunting || (segExpiryAtEnd_ && ends_on_boundary)) ? 1 : 0);
        return limit;
    }
    // A segment has been enqueued.  Its results are needed before the next segment is cut only if a decision can
    // depend on them: a free listener (discover at the boundary) or a time-out that may fire on its last frame.
    // Otherwise - pool full, no time-out in sight: the steady state - up to four more segments are enqueued first
    // (the bank has six sets of buffers), so the device pipeline stays full; the stale LastWrite this leaves in
    // segmentLimit only makes the next cut earlier than necessary, never later.
    int segmentLaunched(int n)
    {
        const int64_t start = framesProcessed_;
        framesProcessed_ += n;
        bool end_decided = false;
        if (speculative_) {
            // while the device runs the spectral half just enqueued: the results of everything older (their listeners
            // ran beside it), so the runes are fed and the clock moves in frame order
            int rc = resolvePending(0);
            if (rc == SDR_OK)
                rc = discoverAhead(start, n, &end_decided);
            if (rc != SDR_OK)
                return rc;
        }
        pending_.push_back(Segment{n, end_decided});
        const bool decide = mode_ == StrainMode && ((listeners_.Available() && !speculative_) || segExpiryAtEnd_);
        if (decide || (int)pending_.size() > 4)
            return resolvePending(decide ? 0 : 4);
        return SDR_OK;
    }
    // The decisions of every cumulation boundary inside the segment [start, start + n), whose spectral half has been
    // enqueued (sdr_defer_listen): for each completed cumulation, in order - the stream clock at that frame, the peaks
    // table's clean-up, Put / FindNext / BindNext / Activate / Attach (:409-426) - with the listener listening from the
    // frame after its boundary (sdr_attach_at).  No time-out can fire inside the segment (segmentLimit), so nothing here
    // depends on what the listeners decode; their runes arrive with the segment's results as usual.  A boundary on
    // the segment's LAST frame is left to afterSegment when a time-out may fire on that frame (it frees a slot first).
    int discoverAhead(int64_t start, int n, bool *end_decided)
    {
        sdr_results r = pollBuffers();
        const auto t0 = std::chrono::steady_clock::now();
        int rc = sdr_poll_peaks(bank_, &r, 1);
        if (rc != SDR_OK)
            return rc;
        const auto t1 = std::chrono::steady_clock::now();
        const int64_t last = start + n - 1;
        for (int ci = 0; ci < r.n_chunks && listeners_.Available(); ci++) {
            const int64_t boundary = r.chunks[ci].frame;
            if (boundary == last && segExpiryAtEnd_)
                break;
            streamClock_.Set(frameTime(boundary));
            reportFrames_ = boundary + 1;
            cleanupPeaks();
            discover(r, ci, boundary + 1);
            if (boundary == last)
                *end_decided = true;
        }
        reportFrames_ = -1;
        const auto t2 = std::chrono::steady_clock::now();
        rc = sdr_process_listen(bank_);
        aheadTiming_[0] += std::chrono::duration<double>(t1 - t0).count();
        aheadTiming_[1] += std::chrono::duration<double>(t2 - t1).count();
        aheadTiming_[2] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t2).count();
        return rc;
    }
    int resolvePending(int keep = 0)
    {
        while ((int)pending_.size() > keep) {
            int64_t ahead = 0;  // frames launched behind the segment being resolved
            for (size_t i = 1; i < pending_.size(); i++)
                ahead += pending_[i].frames;
            reportFrames_ = framesProcessed_ - ahead;
            const int rc = afterSegment(framesProcessed_ - ahead, pending_.front().end_decided);
            reportFrames_ = -1;
            if (rc != SDR_OK)
                return rc;
            pending_.erase(pending_.begin());
        }
        return SDR_OK;
    }
    sdr_results pollBuffers()
    {
        sdr_results r{};
        r.struct_size = sizeof r;
        r.chunks = resChunks_.data();
        r.chunks_cap = (int)resChunks_.size();
        r.peaks = resPeaks_.data();
        r.peaks_cap = (int)resPeaks_.size();
        r.listeners = resListeners_.data();
        r.listeners_cap = (int)resListeners_.size();
        r.edges = resEdges_.data();
        r.edges_cap = (int)resEdges_.size();
        r.runes = resRunes_.data();
        r.rune_frames = resRuneFrames_.data();
        r.runes_cap = (int)resRunes_.size();
        return r;
    }
    // results of the oldest unresolved segment, which ended before frame segment_end (end_decided: discoverAhead has
    // already made the decision of a boundary on its last frame)
    int afterSegment(int64_t segment_end, bool end_decided)
    {
        sdr_results r = pollBuffers();
        const auto ts0 = std::chrono::steady_clock::now();
        const int rc = sdr_poll(bank_, &r, 1);
        if (rc != SDR_OK)
            return rc;
        const auto ts1 = std::chrono::steady_clock::now();
        // runes -> the listeners' text processors, each Write stamped with the time of its frame.  A listener's text
        // processor is independent of every other's (in the reference each runs in a goroutine of its own,
        // rx/text_processor.go:161-171), so listeners are fed in parallel; the reporter calls this triggers are
        // replayed afterwards on this thread, listener by listener.
        std::vector<std::pair<Listener *, const sdr_listener_result *>> work;
        for (int i = 0; i < r.n_listeners; i++) {
            const sdr_listener_result &lr = r.listeners[i];
            if (lr.n_runes == 0)
                continue;
            for (auto &l : listeners_.Listeners())
                if (l->Attached() && l->DeviceID() == lr.listener) {
                    work.emplace_back(l.get(), &lr);
                    break;
                }
        }
        const bool stream_clock = clock_ == &streamClock_;
        auto feed = [&](size_t w) {
            Listener *l = work[w].first;
            const sdr_listener_result &lr = *work[w].second;
            l->BeginFeed();
            for (int k = 0; k < lr.n_runes; k++) {
                if (stream_clock)
                    l->SetFeedTime(frameTime(r.rune_frames[lr.first_rune + k]));
                l->WriteRune(r.runes[lr.first_rune + k]);
            }
            l->EndFeed();
        };
        if (work.size() >= 16 && r.n_runes >= 512) {
            if (!workers_)
                workers_.reset(new Workers((int)std::min(15u, std::max(1u, std::thread::hardware_concurrency()) - 1)));
            workers_->Run(work.size(), feed);
        } else {
            for (size_t w = 0; w < work.size(); w++)
                feed(w);
        }
        const auto ts2 = std::chrono::steady_clock::now();
        for (auto &w : work)
            w.first->FlushEvents();
        const auto ts3 = std::chrono::steady_clock::now();
        segmentTiming_[0] += std::chrono::duration<double>(ts1 - ts0).count();
        segmentTiming_[1] += std::chrono::duration<double>(ts2 - ts1).count();
        segmentTiming_[2] += std::chrono::duration<double>(ts3 - ts2).count();
        // (segments behind this one may already have moved the clock on: discoverAhead.  It only ever runs forward -
        // what is evaluated here, time-outs and the once-a-second clean-up, then sees the later time, as it would a
        // moment later anyway; no listener bound ahead can have timed out by then: segmentLimit)
        if (frameTime(segment_end - 1) > streamClock_.Now() || clock_ != &streamClock_)
            streamClock_.Set(frameTime(segment_end - 1));
        housekeeping();
        checkTimeouts();
        if (mode_ == StrainMode && !end_decided && segment_end % kCumulationSize == 0 && listeners_.Available() && r.n_chunks > 0)
            discover(r, r.n_chunks - 1, segment_end);
        return SDR_OK;
    }
    void housekeeping()  // the cleanupTicker case, :359-363: once per second of clock time
    {
        const double now = clock_->Now();
        if (now - lastCleanup_ >= 1.0) {
            lastCleanup_ = now;
            for (auto &l : listeners_.Listeners())
                l->CheckWriteTimeout();
        }
        cleanupPeaks();
    }
    void cleanupPeaks()  // the peaks table's half of the ticker (discoverAhead runs it boundary by boundary)
    {
        const double now = clock_->Now();
        if (now - lastPeaksCleanup_ < 1.0)
            return;
        lastPeaksCleanup_ = now;
        if (peaks_)
            peaks_->Cleanup();
    }
    void checkTimeouts()  // :396-402, with the clock at the segment's last frame (see Process)
    {
        if (mode_ != StrainMode)
            return;
        std::vector<std::shared_ptr<Listener>> detached;
        for (auto &l : listeners_.Listeners())
            if (l->Attached() && l->TimeoutExceeded()) {
                peaks_->Deactivate(l->GetPeak());
                sdr_detach(bank_, 0, l->DeviceID());
                l->Detach();
                detached.push_back(l);
            }
        for (auto &l : detached)
            listeners_.Release(l);
    }
    // :409-426; the listener hears frame `first_frame` first (the frame after the cumulation's last)
    void discover(const sdr_results &r, int chunk, int64_t first_frame)
    {
        const sdr_chunk_result &cr = r.chunks[chunk];
        lastPeaks_.assign(r.peaks + cr.first_peak, r.peaks + cr.first_peak + cr.n_peaks);
        for (const Peak &p : lastPeaks_)
            peaks_->Put(newPeakCenteredOnSignal(p));
        const Peak *selected = peaks_->FindNext();
        if (!selected)
            return;
        auto listener = listeners_.BindNext();
        if (!listener)
            return;
        const Peak chosen = *selected;
        peaks_->Activate(chosen);
        attach(listener, chosen, first_frame);
    }

    std::string id_;
    ReceiverMode mode_;
    ManualClock streamClock_;
    Clock *clock_;
    std::vector<Reporter *> reporters_;
    Fanout fanout_;
    float peakThreshold_ = SDR_DEFAULT_PEAK_THRESHOLD;
    int edgeWidth_ = SDR_DEFAULT_EDGE_WIDTH;
    int sampleRate_ = 0, blockSize_ = 0;
    int64_t centerFrequency_ = 0, vfoOffset_ = 0;
    double silenceTimeout_ = kDefaultSilenceTimeout, attachmentTimeout_ = kDefaultAttachmentTimeout;
    sdr_bank *bank_ = nullptr;
    std::unique_ptr<PeaksTable> peaks_;
    PeaksTable::RandFn rand_;
    PeaksTable::Policy policy_ = PeaksTable::ReferenceOrder;
    ListenerPool listeners_;
    int64_t framesProcessed_ = 0, reportFrames_ = -1;
    double aheadTiming_[3] = {0, 0, 0}, segmentTiming_[3] = {0, 0, 0};
    int maxBatchFrames_ = 256;
    bool segExpiryAtEnd_ = false;
    bool speculative_ = false;  // the segment being cut is processed spectra first, decisions next, listeners last
    struct Segment {
        int frames;
        bool end_decided;
    };
    std::vector<Segment> pending_;  // the segments enqueued but not yet resolved (oldest first)
    std::unique_ptr<Workers> workers_;
    double lastCleanup_ = 0, lastPeaksCleanup_ = 0;
    std::vector<Peak> lastPeaks_;
    // sdr_poll buffers
    std::vector<sdr_chunk_result> resChunks_;
    std::vector<sdr_peak> resPeaks_;
    std::vector<sdr_listener_result> resListeners_;
    std::vector<sdr_edge> resEdges_;
    std::vector<uint32_t> resRunes_, resRuneFrames_;
};

}  // namespace rx
