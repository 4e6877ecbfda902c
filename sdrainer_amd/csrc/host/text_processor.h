// text_processor.h — C++ host mirror of rx.TextProcessor (rx/text_processor.go): the consumer of the
// runes the GPU decoders emit.  Pure string bookkeeping, host side, as in the reference:
//
//   textWindow              rx/text_processor.go:326-415   (pinned by rx/text_processor_test.go:10-147)
//   callsignExp             rx/text_processor.go:23-25
//   TextProcessor           rx/text_processor.go:57-324    (pinned by :149-179)
//
// Differences, all deliberate: no goroutine / op channel (calls are synchronous, the reference only
// uses the channel to serialise Write against the ticker); the DXCC and Supercheck finders, which the
// reference downloads in its constructor (:94-136), are optional interfaces that default to absent
// (then every syntactically valid call passes and carries weight 0, as in the reference when its
// finders are nil); `bestMatch` breaks ties in key order where the reference iterates a Go map
// (unspecified order).  callsign.Parse comes from github.com/ftl/hamradio v0.2.9 (not vendored): its
// published syntax (optional prefix "/", base call, optional suffix, optional working condition) is
// restated here.
#pragma once
#include <algorithm>
#include <cctype>
#include <functional>
#include <map>
#include <regex>
#include <string>
#include <vector>

namespace rx {

constexpr int kDefaultTextWindowSize = 20;  // :17
constexpr int kSpottingThreshold = 3;       // :18
constexpr double kDefaultWriteTimeout = 5;  // :20 (seconds)

inline const std::regex &callsignExp()  // :23-25
{
    static const std::regex re(
        R"(\s(?:([a-z0-9]+)/)?(([a-z]|[a-z][a-z]|[0-9][a-z]|[0-9][a-z][a-z])[0-9][a-z0-9]*[a-z])(?:/([a-z0-9]+))?(?:/(p|a|m|mm|am))?)",
        std::regex::ECMAScript);
    return re;
}

class TextWindow {  // :326-415
public:
    explicit TextWindow(int windowSize) : windowSize_(windowSize) {}
    const std::string &String() const { return window_[current_]; }
    void Reset()
    {
        window_[0].clear();
        window_[1].clear();
        current_ = 0;  // (searchPoint is not reset by the reference either)
    }
    // returns the number of bytes taken; -1 = "text window is full, use Shift() before writing again"
    int Write(const std::string &bytes)
    {
        const int appendLen = std::min((int)bytes.size(), windowSize_ - (int)window_[current_].size());
        if (!bytes.empty() && appendLen == 0)
            return -1;
        window_[current_].append(bytes, 0, (size_t)appendLen);
        return appendLen;
    }
    void Shift()
    {
        const int other = (current_ + 1) % 2;
        const int halfSize = windowSize_ / 2;
        window_[other].clear();
        const std::string &cur = window_[current_];
        const int startIndex = std::max(0, (int)cur.size() - halfSize);
        const int appendLen = std::min(halfSize, (int)cur.size() - startIndex);
        if (appendLen > 0)
            window_[other].append(cur, (size_t)startIndex, (size_t)appendLen);
        current_ = other;
        searchPoint_ = std::max(0, searchPoint_ - startIndex);
    }
    bool IsFull() const { return (int)window_[current_].size() == windowSize_; }
    bool FindNext(const std::regex &exp, bool includeTail, std::string *out)
    {
        const std::string &cur = window_[current_];
        if (searchPoint_ >= (int)cur.size())
            return false;
        std::smatch m;
        const std::string searchText = cur.substr((size_t)searchPoint_);
        if (!std::regex_search(searchText, m, exp))
            return false;
        const int end = (int)(m.position(0) + m.length(0));
        if (!includeTail && end >= (int)searchText.size())
            return false;
        searchPoint_ += end;
        *out = m.str(0);
        return true;
    }
    int CurrentWindow() const { return current_; }
    int SearchPoint() const { return searchPoint_; }
    void Preset(const std::string &s) { window_[current_] = s; }  // tests only

private:
    std::string window_[2];
    int windowSize_;
    int current_ = 0;
    int searchPoint_ = 0;
};

// hamradio/callsign.Parse + Callsign.String(): PREFIX/BASE/SUFFIX/WORKING_CONDITION, upper case
inline bool ParseCallsign(const std::string &s, std::string *canonical)
{
    static const std::regex re(
        R"(^(?:([A-Z0-9]+)/)?((?:[A-Z]|[A-Z][A-Z]|[0-9][A-Z]|[0-9][A-Z][A-Z])[0-9][A-Z0-9]*[A-Z])(?:/([A-Z0-9]+))?(?:/(P|A|M|MM|AM))?$)",
        std::regex::ECMAScript);
    std::string up;
    for (char c : s)
        up.push_back((char)std::toupper((unsigned char)c));
    std::smatch m;
    if (!std::regex_match(up, m, re))
        return false;
    std::string out;
    if (m[1].matched)
        out += m[1].str() + "/";
    out += m[2].str();
    if (m[3].matched)
        out += "/" + m[3].str();
    if (m[4].matched)
        out += "/" + m[4].str();
    *canonical = out;
    return true;
}

struct CallsignReporter {  // :27-31
    virtual ~CallsignReporter() = default;
    virtual void CallsignDecoded(const std::string &callsign, int count, int weight) = 0;
    virtual void CallsignSpotted(const std::string &callsign) = 0;
    virtual void SpotTimeout(const std::string &callsign) = 0;
};

class TextProcessor {  // :57-324
public:
    using Finder = std::function<bool(const std::string &)>;
    TextProcessor(std::function<double()> now, CallsignReporter *reporter)
        : now_(std::move(now)), reporter_(reporter), lastWrite_(now_()), window_(kDefaultTextWindowSize)
    {
    }
    void SetDXCCFinder(Finder f) { dxcc_ = std::move(f); }  // absent = every call is valid (:297-303)
    void SetSCPFinder(Finder f) { scp_ = std::move(f); }    // absent = weight 0 (:321-324)

    void Restart()  // :175-182
    {
        lastWrite_ = now_();
        lastBestMatch_.clear();
        window_.Reset();
        collected_.clear();
    }
    double LastWrite() const { return lastWrite_; }
    void CheckWriteTimeout()  // :188-193
    {
        if (now_() - lastWrite_ > kDefaultWriteTimeout)
            WriteTimeout();
    }
    void WriteTimeout()  // :195-200
    {
        std::string candidate;
        if (window_.FindNext(callsignExp(), true, &candidate))
            collectCallsign(candidate);
    }
    void Write(const std::string &bytes)  // :202-216 + findNextCallsign :218-242
    {
        lastWrite_ = now_();
        std::string rest = bytes;
        while (!rest.empty()) {
            const int n = window_.Write(rest);
            if (n < 0)
                break;  // the reference panics here; cannot happen because a full window is shifted below
            std::string candidate;
            if (window_.FindNext(callsignExp(), false, &candidate))
                collectCallsign(candidate);
            if (n <= (int)rest.size())
                rest = rest.substr((size_t)n);
            if (window_.IsFull())
                window_.Shift();
        }
    }
    int Count(const std::string &call) const
    {
        auto it = collected_.find(call);
        return it == collected_.end() ? 0 : it->second.count;
    }
    const TextWindow &Window() const { return window_; }

private:
    struct Collected {
        int weight = 0, count = 0;
    };
    void collectCallsign(std::string candidate)  // :244-280
    {
        // strings.ToLower(strings.TrimSpace(candidate))
        size_t a = 0, b = candidate.size();
        while (a < b && std::isspace((unsigned char)candidate[a]))
            a++;
        while (b > a && std::isspace((unsigned char)candidate[b - 1]))
            b--;
        candidate = candidate.substr(a, b - a);
        for (char &c : candidate)
            c = (char)std::tolower((unsigned char)c);
        if (candidate.rfind("tu5nn", 0) == 0)  // isFalsePositive :282-295
            return;
        std::string call;
        if (!ParseCallsign(candidate, &call))
            return;
        if (dxcc_ && !dxcc_(call))
            return;
        auto it = collected_.find(call);
        if (it == collected_.end())
            it = collected_.emplace(call, Collected{(scp_ && scp_(call)) ? 1 : 0, 0}).first;
        it->second.count++;
        if (reporter_)
            reporter_->CallsignDecoded(call, it->second.count, it->second.weight);
        // bestMatch :305-319
        std::string best;
        int maxCount = kSpottingThreshold - 1;
        for (const auto &kv : collected_) {
            const int weighted = kv.second.count + kv.second.weight;
            if (maxCount < weighted) {
                maxCount = weighted;
                best = kv.first;
            }
        }
        if (best.empty())
            return;
        if (best != lastBestMatch_ && !lastBestMatch_.empty() && reporter_)
            reporter_->SpotTimeout(lastBestMatch_);
        if (reporter_)
            reporter_->CallsignSpotted(best);
        lastBestMatch_ = best;
    }

    std::function<double()> now_;
    CallsignReporter *reporter_;
    double lastWrite_;
    std::string lastBestMatch_;
    TextWindow window_;
    std::map<std::string, Collected> collected_;
    Finder dxcc_, scp_;
};

}  // namespace rx
