// Tests of the C++ host mirror (sdrainer_amd/csrc/host/rx.h), modelled on the reference's
// rx/peaks_test.go and rx/listener_test.go.  `test_rx_host cpu` needs no GPU; `test_rx_host strain
// <iq.f32> <rate> <N> <frames> <pool>` drives a strain-mode Receiver on the GPU and prints what it did
// as JSON for tests/test_host_mirror.py to compare with an oracle-driven simulation.
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../sdrainer_amd/csrc/host/rx.h"

#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) {                                                           \
            fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);    \
            failures++;                                                          \
        }                                                                        \
    } while (0)

static int failures = 0;

static rx::Peak mk(int from, int to)
{
    rx::Peak p{};
    p.from = from;
    p.to = to;
    return p;
}

static void TestIDPool()  // rx/listener_test.go:10-35
{
    rx::IDPool p(10, "test");
    for (int i = 1; i <= 10; i++) {
        std::string id;
        CHECK(p.Pop(&id));
        CHECK(id == "test" + std::to_string(i));
    }
    std::string id;
    CHECK(!p.Pop(&id));
    for (int i = 1; i <= 10; i++)
        p.Push("test" + std::to_string(i));
    CHECK(p.Len() == 10);
    p.Push("one more");
    CHECK(p.Len() == 11);
    CHECK(p.Pop(&id) && id == "one more");
}

static void TestListenerPool()  // rx/listener_test.go:37-67
{
    rx::ManualClock clock;
    rx::ListenerPool pool(3, "test", [&](const std::string &id) { return std::make_shared<rx::Listener>(id, &clock, nullptr); });
    std::vector<std::shared_ptr<rx::Listener>> ls;
    for (int i = 1; i <= 3; i++) {
        auto l = pool.BindNext();
        CHECK(l != nullptr);
        ls.push_back(l);
        CHECK(pool.Listeners()[i - 1] == l);
        CHECK(l->ID() == "test" + std::to_string(i));
    }
    CHECK(pool.BindNext() == nullptr);
    pool.Release(ls[1]);
    CHECK(pool.Listeners().size() == 2 && pool.Listeners()[0] == ls[0] && pool.Listeners()[1] == ls[2]);
    pool.Release(ls[0]);
    CHECK(pool.Listeners().size() == 1 && pool.Listeners()[0] == ls[2]);
    pool.Release(ls[2]);
    CHECK(pool.Listeners().empty());
    auto n = pool.BindNext();
    CHECK(n && n->ID() == ls[2]->ID());
}

static void TestPeaksTable_PutIntoEmptyTable()  // rx/peaks_test.go:12-26
{
    rx::ManualClock clock;
    clock.Set(5);
    rx::PeaksTable t(512, &clock);
    t.Put(mk(234, 235));
    CHECK(t.GetEntry(234) && t.GetEntry(234) == t.GetEntry(235));
    CHECK(t.GetEntry(234)->state == rx::peakNew && t.GetEntry(234)->since == 5);
    CHECK(t.Get(233) == nullptr && t.Get(-1) == nullptr && t.Get(512) == nullptr);
}

static void TestPeaksTable_Put()  // rx/peaks_test.go:28-72
{
    rx::ManualClock clock;
    rx::PeaksTable t(12, &clock);
    t.Put(mk(3, 4));
    t.Put(mk(5, 6));
    t.Put(mk(8, 8));
    t.Activate(mk(8, 8));
    t.Put(mk(10, 10));
    t.Activate(mk(10, 10));
    t.Deactivate(mk(10, 10));
    const rx::PeaksTable::Entry *p3 = t.GetEntry(8), *p4 = t.GetEntry(10);
    CHECK(p3->state == rx::peakActive && p4->state == rx::peakInactive);
    t.Put(mk(1, 2));
    t.Put(mk(4, 5));
    t.Put(mk(7, 8));
    t.Put(mk(10, 11));
    CHECK(t.GetEntry(0) == nullptr);
    CHECK(t.Get(1) && t.Get(1)->from == 1 && t.Get(2)->to == 2);
    CHECK(t.GetEntry(3) == nullptr);
    CHECK(t.Get(4) && t.Get(4)->from == 4 && t.Get(5)->to == 5);
    CHECK(t.GetEntry(6) == nullptr && t.GetEntry(7) == nullptr);
    CHECK(t.GetEntry(8) == p3);
    CHECK(t.GetEntry(9) == nullptr);
    CHECK(t.GetEntry(10) == p4);
    CHECK(t.GetEntry(11) == nullptr);
}

static void TestPeaksTable_Cleanup()  // rx/peaks_test.go:74-124
{
    rx::ManualClock clock;
    clock.Set(1000);
    {
        rx::PeaksTable t(512, &clock);
        t.Put(mk(234, 235));
        t.Cleanup();
        CHECK(t.Get(234) && t.Get(235));
        clock.Add(rx::kDefaultPeakTimeout + 1);
        t.Cleanup();
        CHECK(!t.Get(234) && !t.Get(235));
    }
    {
        rx::PeaksTable t(512, &clock);
        t.Put(mk(234, 235));
        t.Cleanup();
        t.Activate(mk(234, 235));
        clock.Add(rx::kDefaultPeakTimeout + 1);
        t.Cleanup();
        CHECK(t.Get(234) && t.Get(235));
        t.Deactivate(mk(234, 235));
        t.Cleanup();
        CHECK(!t.Get(234) && !t.Get(235));
    }
}

static void TestPeaksTable_FindNext()  // rx/peaks_test.go:126-143
{
    rx::ManualClock clock;
    rx::PeaksTable t(512, &clock);
    unsigned seed = 12345;
    t.SetRand([&](int n) { seed = seed * 1103515245u + 12345u; return (int)((seed >> 8) % (unsigned)n); });
    t.Put(mk(234, 235));
    const rx::Peak *next = t.FindNext();
    CHECK(next && next->from == 234 && next->to == 235);
    t.Activate(*next);
    CHECK(t.FindNext() == nullptr);
    t.Deactivate(mk(234, 235));
    CHECK(t.FindNext() == nullptr);
}

static void TestPeaksTable_StrongestFirst()  // reproducible selection without a seed
{
    rx::ManualClock clock;
    rx::PeaksTable t(512, &clock);
    t.SetPolicy(rx::PeaksTable::StrongestFirst);
    rx::Peak a = mk(10, 12), b = mk(100, 101), c = mk(200, 205), d = mk(300, 300);
    a.signal_value = 20.f;
    b.signal_value = 35.f;
    c.signal_value = 35.f;  // tie with b: the lower bin wins
    d.signal_value = 50.f;
    t.Put(a);
    t.Put(b);
    t.Put(c);
    t.Put(d);
    const rx::Peak *n = t.FindNext();
    CHECK(n && n->from == 300);
    t.Activate(*n);
    n = t.FindNext();
    CHECK(n && n->from == 100);
    t.Activate(*n);
    n = t.FindNext();
    CHECK(n && n->from == 200);
    t.Activate(*n);
    n = t.FindNext();
    CHECK(n && n->from == 10);
    t.Activate(*n);
    CHECK(t.FindNext() == nullptr);
}

static void TestListenerTimeouts()  // rx/listener.go:126-136
{
    rx::ManualClock clock;
    rx::Listener l("a", &clock, nullptr);
    l.SetSilenceTimeout(20);
    l.SetAttachmentTimeout(120);
    l.Attach(mk(5, 5), 0);
    CHECK(!l.TimeoutExceeded());
    clock.Add(19);
    l.Write("e");
    clock.Add(19);
    CHECK(!l.TimeoutExceeded());
    clock.Add(2);
    CHECK(l.TimeoutExceeded());  // silence
    l.Write("t");
    CHECK(!l.TimeoutExceeded());
    clock.Set(121);
    l.Write("t");
    CHECK(l.TimeoutExceeded());  // attachment
}

static void TestPeakCentering()  // rx/receiver.go:474-500 + dsp/fft_test.go:31-50
{
    rx::Receiver r("rx", rx::StrainMode);
    r.SetCenterFrequency(7020000);
    // geometry without a bank: use the mapping helpers through a started-less receiver is not possible,
    // so check the mapping directly
    host::FrequencyMapping m(48000, 512, 7020000);
    CHECK(m.FrequencyToBin(7020000 - 24000) == 0 && m.BinToFrequency(0, host::BinCenter) == 7020000 - 24000);
    CHECK(m.FrequencyToBin(7020000) == 256 && m.BinToFrequency(256, host::BinCenter) == 7020000);
}

struct PrintReporter : rx::Reporter {
    std::vector<std::string> events;
    std::vector<long long> event_frames;  // frames processed when the event was reported
    rx::Receiver *receiver = nullptr;
    void stamp() { event_frames.push_back(receiver ? (long long)receiver->FramesProcessed() : -1); }
    void ListenerActivated(const std::string &l, int64_t f) override
    {
        events.push_back("+" + l + "@" + std::to_string(f));
        stamp();
    }
    void ListenerDeactivated(const std::string &l, int64_t f) override
    {
        events.push_back("-" + l + "@" + std::to_string(f));
        stamp();
    }
    // rx/rx.go:14-16
    std::vector<std::string> callsigns;
    void CallsignDecoded(const std::string &l, const std::string &c, int64_t f, int count, int weight) override
    {
        callsigns.push_back(l + " decoded " + c + " " + std::to_string(f) + " " + std::to_string(count) + " " + std::to_string(weight));
    }
    void CallsignSpotted(const std::string &l, const std::string &c, int64_t f) override
    {
        callsigns.push_back(l + " spotted " + c + " " + std::to_string(f));
    }
    void SpotTimeout(const std::string &l, const std::string &c, int64_t f) override
    {
        callsigns.push_back(l + " timeout " + c + " " + std::to_string(f));
    }
};

static int run_strain(const char *path, int rate, int n, int frames, int pool, bool strongest = false, double silence = 1e9,
                      double attachment = 1e9, int max_batch = 256, int piece = 0)
{
    FILE *f = fopen(path, "rb");
    if (!f)
        return 2;
    std::vector<float> iq((size_t)frames * 2 * n);
    if (fread(iq.data(), sizeof(float), iq.size(), f) != iq.size())
        return 2;
    fclose(f);
    PrintReporter rep;  // must outlive the receiver: Stop() reports the final deactivations
    rx::Receiver r("rx", rx::StrainMode, nullptr, pool);
    rep.receiver = &r;
    r.AddReporter(&rep);
    r.SetCenterFrequency(7020000);
    if (strongest)
        r.SetSelectionPolicy(rx::PeaksTable::StrongestFirst);
    r.SetSilenceTimeout(silence);
    r.SetAttachmentTimeout(attachment);
    r.SetEdgeWidth(70 * n / 512);
    if (r.Start(rate, n, max_batch) != SDR_OK) {
        fprintf(stderr, "Start failed: %s\n", sdr_last_error());
        return 3;
    }
    // error behaviour of IQData (rx/receiver.go:315-334)
    if (r.IQData(rate + 1, iq.data(), 2 * (size_t)n) != SDR_ERR_BAD_RATE || r.IQData(rate, iq.data(), 2 * (size_t)n - 2) != SDR_ERR_BAD_SIZE)
        return 4;
    // frames arrive in ragged pieces, Process() is called whenever some are staged
    // (or, argv[11], in equal pieces of that many frames: long segments with many cumulation boundaries each)
    const int pieces[] = {1, 37, 100, 163, 7, 250};
    int done = 0, k = 0;
    while (done < frames) {
        const int m = std::min(piece > 0 ? piece : pieces[k++ % 6], frames - done);
        if (r.IQData(rate, iq.data() + (size_t)done * 2 * n, (size_t)m * 2 * n) != SDR_OK)
            return 5;
        done += m;
        if (r.Process() != SDR_OK) {
            fprintf(stderr, "Process failed: %s\n", sdr_last_error());
            return 6;
        }
    }
    printf("{\"frames\": %lld, \"events\": [", (long long)r.FramesProcessed());
    for (size_t i = 0; i < rep.events.size(); i++)
        printf("%s\"%s\"", i ? ", " : "", rep.events[i].c_str());
    printf("], \"event_frames\": [");
    for (size_t i = 0; i < rep.event_frames.size(); i++)
        printf("%s%lld", i ? ", " : "", rep.event_frames[i]);
    printf("], \"callsigns\": [");
    for (size_t i = 0; i < rep.callsigns.size(); i++)
        printf("%s\"%s\"", i ? ", " : "", rep.callsigns[i].c_str());
    printf("], \"listeners\": [");
    bool first = true;
    for (auto &l : r.Listeners().Listeners()) {
        printf("%s{\"id\": \"%s\", \"bin\": %d, \"frequency\": %lld, \"text\": \"", first ? "" : ", ", l->ID().c_str(), l->SignalBin(),
               (long long)l->GetPeak().signal_frequency);
        for (unsigned char c : l->Text())
            printf("\\u%04x", c);  // raw UTF-8 bytes, re-assembled by the test
        printf("\"}");
        first = false;
    }
    printf("]}\n");
    return 0;
}


// DecodeMode (rx/receiver.go:272-297): SetVFOOffset forces a peak at the VFO frequency and the single listener
// decodes it; a second SetVFOOffset moves the listener.  Prints the listener's events and text as JSON.
static int run_decode(const char *path, int rate, int n, int frames, long long vfo_offset)
{
    FILE *f = fopen(path, "rb");
    if (!f)
        return 2;
    std::vector<float> iq((size_t)frames * 2 * n);
    if (fread(iq.data(), sizeof(float), iq.size(), f) != iq.size())
        return 2;
    fclose(f);
    PrintReporter rep;
    rx::Receiver r("rx", rx::DecodeMode);
    r.AddReporter(&rep);
    r.SetCenterFrequency(7020000);
    r.SetEdgeWidth(70 * n / 512);
    if (r.Start(rate, n, 256) != SDR_OK)
        return 3;
    if (r.SetVFOOffset(vfo_offset) != SDR_OK)
        return 4;
    const int half = frames / 2;
    auto feed = [&](int from, int to) {
        for (int f = from; f < to; f += 200) {
            const int m = std::min(200, to - f);
            if (r.IQData(rate, iq.data() + (size_t)f * 2 * n, (size_t)m * 2 * n) != SDR_OK || r.Process() != SDR_OK)
                return false;
        }
        return true;
    };
    if (!feed(0, half))
        return 5;
    const auto l0 = r.Listeners().First();
    const std::string text0 = l0 ? l0->Text() : "";
    const int bin0 = l0 ? l0->SignalBin() : -1;
    // retune to the same offset: the pool of one is reset and a fresh listener (new decoder) takes over
    if (r.SetVFOOffset(vfo_offset) != SDR_OK)
        return 6;
    if (!feed(half, frames))
        return 7;
    const auto l1 = r.Listeners().First();
    printf("{\"frames\": %lld, \"bin\": %d, \"events\": [", (long long)r.FramesProcessed(), bin0);
    for (size_t i = 0; i < rep.events.size(); i++)
        printf("%s\"%s\"", i ? ", " : "", rep.events[i].c_str());
    printf("], \"text0\": \"");
    for (unsigned char c : text0)
        printf("\\u%04x", c);
    printf("\", \"text1\": \"");
    for (unsigned char c : (l1 ? l1->Text() : std::string()))
        printf("\\u%04x", c);
    printf("\", \"peaks_found\": %d}\n", (int)r.LastPeaks().size());
    return 0;
}

// ---- rx/text_processor_test.go ---------------------------------------------------------------
static void TestTextWindow_Write()  // :10-69
{
    struct Case {
        const char *preset, *text, *expected;
        int expectedN;
        bool invalid;
    } tt[] = {
        {"", "", "", 0, false},
        {"", "abc", "abc", 3, false},
        {"123", "abc", "123abc", 3, false},
        {"1234567", "abcdef", "1234567abc", 3, false},
        {"1234567890", "abcdef", "1234567890", 0, true},
    };
    for (const Case &tc : tt) {
        rx::TextWindow w(10);
        w.Preset(tc.preset);
        const int n = w.Write(tc.text);
        CHECK((n < 0) == tc.invalid);
        CHECK(std::max(n, 0) == tc.expectedN);
        CHECK(w.String() == tc.expected);
    }
}

static void TestTextWindow_Shift()  // :71-105
{
    rx::TextWindow w(10);
    w.Shift();
    CHECK(w.CurrentWindow() == 1 && w.String() == "");
    CHECK(w.Write("1234") >= 0);
    w.Shift();
    CHECK(w.CurrentWindow() == 0 && w.String() == "1234");
    CHECK(w.Write("123456") >= 0);
    w.Shift();
    CHECK(w.CurrentWindow() == 1 && w.String() == "23456");
    CHECK(w.Write("abcdefg") >= 0);
    w.Shift();
    CHECK(w.CurrentWindow() == 0 && w.String() == "abcde");
    CHECK(w.Write("fg") >= 0);
    w.Shift();
    CHECK(w.CurrentWindow() == 1 && w.String() == "cdefg");
    w.Reset();
    CHECK(w.CurrentWindow() == 0 && w.String() == "");
}

static void TestTextWindow_FindNext()  // :107-135
{
    rx::TextWindow w(10);
    const std::regex aExp("a");
    std::string out;
    CHECK(!w.FindNext(aExp, true, &out));
    CHECK(w.SearchPoint() == 0);
    w.Write("abc");
    CHECK(w.FindNext(aExp, true, &out));
    CHECK(w.SearchPoint() == 1);
    CHECK(!w.FindNext(aExp, true, &out));
    CHECK(w.SearchPoint() == 1);
    w.Write("1234567");
    w.Shift();
    CHECK(w.SearchPoint() == 0);
    CHECK(w.String() == "34567");
    w.Write("abc");
    CHECK(w.FindNext(aExp, true, &out));
    CHECK(w.SearchPoint() == 6);
    w.Shift();
    CHECK(w.SearchPoint() == 3);
    CHECK(w.String() == "67abc");
}

static void TestTextWindow_FindNext_IncludeTail()  // :137-147
{
    rx::TextWindow w(10);
    const std::regex abcExp("abc");
    std::string out;
    w.Write("12345abc");
    CHECK(!w.FindNext(abcExp, false, &out));
    CHECK(w.FindNext(abcExp, true, &out));
    CHECK(out == "abc");
}

static void TestTextProcessor_CollectCallsign()  // :149-162
{
    rx::TextProcessor p([] { return 0.0; }, nullptr);
    for (char c : std::string("cq cq cq de dl1abc dl1abc dl1abc pse k"))
        p.Write(std::string(1, c));
    CHECK(p.Count("DL1ABC") == 3);
}

static void TestTextProcessor_WriteTimeout()  // :164-179
{
    rx::TextProcessor p([] { return 0.0; }, nullptr);
    for (char c : std::string("cq de dl1abc"))
        p.Write(std::string(1, c));
    CHECK(p.Count("DL1ABC") == 0);
    p.WriteTimeout();
    CHECK(p.Count("DL1ABC") == 1);
}

// `test_rx_host text`: replay a script from stdin through a TextProcessor and print its reporter
// events, one per line.  Script lines: "W <text>" (written rune by rune), "B <text>" (one Write),
// "A <seconds>" (advance the clock, then CheckWriteTimeout as the receiver's ticker does), "R" (Restart).
// CallsignSearch (straight-line code) against callsignExp (std::regex) on random windows: same answer - found or not,
// where, how long - for every one of them, and the two FindNext variants walk a window identically.  argv: how many.
static void TestCallsignSearchAgainstTheRegex(long count)
{
    // mostly the decoder's alphabet, weighted towards what makes and breaks call signs; some bytes the regex must skip
    static const char alphabet[] = "aaabcdklmpmmz001299  //  \t-.?\xc3\xa4\xc2\xa6\n";
    const int na = (int)sizeof(alphabet) - 1;
    unsigned long long x = 88172645463325252ull;
    auto rnd = [&] {
        x ^= x << 13;
        x ^= x >> 7;
        x ^= x << 17;
        return x;
    };
    const std::regex &re = rx::callsignExp();
    long found = 0;
    for (long it = 0; it < count; it++) {
        const int n = (int)(rnd() % 25);
        std::string w;
        for (int i = 0; i < n; i++)
            w.push_back(alphabet[rnd() % na]);
        if (it % 3 == 0 && n > 8) {  // plant something call-shaped
            static const char *seeds[] = {" dl1abc", " 9a1aa/p", " ea8/dl1abc/mm", " w1aw ", " 1a2b/am", " k1a/7 ", " 2e0abc/m"};
            const char *sd = seeds[rnd() % 7];
            w.replace(rnd() % (unsigned)(n - 7), std::min<size_t>(strlen(sd), 7), sd);
        }
        std::smatch m;
        const bool a = std::regex_search(w, m, re);
        int pos = -1, len = -1;
        const bool b = rx::CallsignSearch(w.data(), (int)w.size(), &pos, &len);
        if (a != b || (a && (m.position(0) != pos || m.length(0) != len))) {
            failures++;
            fprintf(stderr, "CallsignSearch differs from callsignExp on \"%s\": regex %d (%ld, %ld), code %d (%d, %d)\n", w.c_str(), (int)a,
                    a ? (long)m.position(0) : -1L, a ? (long)m.length(0) : -1L, (int)b, pos, len);
            if (failures > 10)
                return;
        }
        found += a;
        // the two FindNext variants on the same window, call after call
        rx::TextWindow w1(24), w2(24);
        w1.Preset(w);
        w2.Preset(w);
        for (int k = 0; k < 4; k++) {
            std::string o1, o2;
            const bool tail = (k & 1) != 0;
            const bool f1 = w1.FindNext(re, tail, &o1), f2 = w2.FindNextCallsign(tail, &o2);
            if (f1 != f2 || o1 != o2 || w1.SearchPoint() != w2.SearchPoint()) {
                failures++;
                fprintf(stderr, "FindNextCallsign differs from FindNext on \"%s\" (call %d)\n", w.c_str(), k);
                break;
            }
        }
    }
    CHECK(found > count / 20);  // (the generator does produce matches)

    // callsign.Parse's syntax: the straight-line check against the anchored regex, and the canonical form against the
    // parts the regex captures, joined by slashes
    static const char up_alphabet[] = "AABDKLMMPZ00129//";
    const int nu = (int)sizeof(up_alphabet) - 1;
    long ok = 0;
    for (long it = 0; it < count; it++) {
        const int n = 1 + (int)(rnd() % 14);
        std::string w;
        for (int i = 0; i < n; i++)
            w.push_back(up_alphabet[rnd() % nu]);
        if (it % 4 == 0) {
            static const char *seeds[] = {"DL1ABC", "9A1AA/P", "EA8/DL1ABC/MM", "W1AW", "1A2B/AM", "K1A/7", "2E0ABC/M", "F/DL1ABC/P", "DL1ABC/P/M", "M/1A2B/A"};
            w = seeds[rnd() % 10];
            if (rnd() % 3 == 0)
                w[rnd() % w.size()] = up_alphabet[rnd() % nu];
        }
        std::smatch m;
        const bool a = std::regex_match(w, m, rx::callsignSyntax());
        std::string canon;
        const bool b = rx::ParseCallsign(w, &canon);
        std::string want;
        if (a) {
            if (m[1].matched)
                want += m[1].str() + "/";
            want += m[2].str();
            if (m[3].matched)
                want += "/" + m[3].str();
            if (m[4].matched)
                want += "/" + m[4].str();
        }
        if (a != b || (a && canon != want)) {
            failures++;
            fprintf(stderr, "ParseCallsign differs from the syntax regex on \"%s\": regex %d \"%s\", code %d \"%s\"\n", w.c_str(), (int)a, want.c_str(), (int)b,
                    canon.c_str());
            if (failures > 10)
                return;
        }
        ok += a;
    }
    CHECK(ok > count / 50);
}

// rx::Workers: every item of every run exactly once, whatever the threads' timing (runs back to back, so threads woken
// for one run meet the next; also run under -fsanitize=thread by tests/test_host_mirror.py when the compiler has it)
static void TestWorkers()
{
    rx::Workers pool(7);
    std::vector<std::atomic<int>> hits(1000);
    long total = 0;
    for (int run = 0; run < 3000; run++) {
        const size_t n = 1 + (size_t)(run * 7919 % 997);
        for (size_t i = 0; i < n; i++)
            hits[i].store(0, std::memory_order_relaxed);
        std::atomic<long> sum{0};
        pool.Run(n, [&](size_t i) {
            hits[i].fetch_add(1, std::memory_order_relaxed);
            sum.fetch_add((long)i, std::memory_order_relaxed);
        });
        bool once = true;
        for (size_t i = 0; i < n; i++)
            once = once && hits[i].load(std::memory_order_relaxed) == 1;
        CHECK(once);
        CHECK(sum.load() == (long)(n * (n - 1) / 2));
        total += (long)n;
    }
    CHECK(total > 0);
}

static int run_text()
{
    struct Printer : rx::CallsignReporter {
        void CallsignDecoded(const std::string &c, int count, int weight) override { printf("decoded %s %d %d\n", c.c_str(), count, weight); }
        void CallsignSpotted(const std::string &c) override { printf("spotted %s\n", c.c_str()); }
        void SpotTimeout(const std::string &c) override { printf("timeout %s\n", c.c_str()); }
    } printer;
    double now = 0;
    rx::TextProcessor p([&now] { return now; }, &printer);
    char line[65536];
    while (fgets(line, sizeof line, stdin)) {
        std::string s(line);
        while (!s.empty() && (s.back() == '\n' || s.back() == '\r'))
            s.pop_back();
        if (s.empty())
            continue;
        const std::string arg = s.size() > 2 ? s.substr(2) : "";
        switch (s[0]) {
        case 'W':
            for (char c : arg)
                p.Write(std::string(1, c));
            break;
        case 'B':
            p.Write(arg);
            break;
        case 'A':
            now += atof(arg.c_str());
            p.CheckWriteTimeout();
            break;
        case 'R':
            p.Restart();
            break;
        }
    }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc >= 2 && !strcmp(argv[1], "cpu")) {
        TestIDPool();
        TestListenerPool();
        TestPeaksTable_PutIntoEmptyTable();
        TestPeaksTable_Put();
        TestPeaksTable_Cleanup();
        TestPeaksTable_FindNext();
        TestPeaksTable_StrongestFirst();
        TestListenerTimeouts();
        TestPeakCentering();
        TestTextWindow_Write();
        TestTextWindow_Shift();
        TestTextWindow_FindNext();
        TestTextWindow_FindNext_IncludeTail();
        TestTextProcessor_CollectCallsign();
        TestTextProcessor_WriteTimeout();
        TestCallsignSearchAgainstTheRegex(argc >= 3 ? atol(argv[2]) : 200000);
        TestWorkers();
        printf("%s\n", failures ? "FAILED" : "ok");
        return failures ? 1 : 0;
    }
    if (argc >= 2 && !strcmp(argv[1], "text"))
        return run_text();
    if (argc >= 7 && !strcmp(argv[1], "strain"))
        return run_strain(argv[2], atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]),
                          argc >= 8 && !strcmp(argv[7], "strongest"), argc >= 9 ? atof(argv[8]) : 1e9,
                          argc >= 10 ? atof(argv[9]) : 1e9, argc >= 11 ? atoi(argv[10]) : 256, argc >= 12 ? atoi(argv[11]) : 0);
    if (argc >= 7 && !strcmp(argv[1], "decode"))
        return run_decode(argv[2], atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoll(argv[6]));
    fprintf(stderr, "usage: %s cpu | text | strain <iq.f32> <rate> <N> <frames> <pool>\n", argv[0]);
    return 2;
}
