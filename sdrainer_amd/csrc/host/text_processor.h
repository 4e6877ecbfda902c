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

// callsignExp as straight-line code: the leftmost match in the order a backtracking matcher (Go's regexp for this
// pattern, std::regex ECMAScript) finds it - first whitespace position from which the pattern matches; the prefix
// "xx/" tried before no prefix; the four shapes of the call's head in the order they are written; the greedy body
// backed off to its last letter; then, nothing mandatory being left, each optional suffix taken greedily.  std::regex
// needs a microsecond or two per 20-character window and runs once per decoded rune per listener (6 000 times per
// 2048-frame batch at 256 listeners): it was what bounded the end-to-end rate with a full pool.  Checked against
// callsignExp itself on millions of random windows (tests/host/test_rx_host.cpp "cpu").
inline bool CallsignSearch(const char *s, int n, int *pos, int *len)
{
    auto space = [](char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\f' || c == '\r'; };  // Go RE2's \\s is [\\t\\n\\f\\r ] (rx/text_processor.go:24): no \\v
    auto letter = [](char c) { return c >= 'a' && c <= 'z'; };
    auto digit = [](char c) { return c >= '0' && c <= '9'; };
    auto alnum = [&](char c) { return letter(c) || digit(c); };
    auto at = [&](int i) { return i < n ? s[i] : '\0'; };
    auto run = [&](int i) {  // length of the [a-z0-9] run starting at i
        int k = 0;
        while (i + k < n && alnum(s[i + k]))
            k++;
        return k;
    };
    // (([a-z]|[a-z][a-z]|[0-9][a-z]|[0-9][a-z][a-z])[0-9][a-z0-9]*[a-z]) then the optional suffixes; returns the end or -1
    auto call_from = [&](int q) -> int {
        for (int alt = 0; alt < 4; alt++) {
            int h;  // end of the head
            if (alt == 0 && letter(at(q)))
                h = q + 1;
            else if (alt == 1 && letter(at(q)) && letter(at(q + 1)))
                h = q + 2;
            else if (alt == 2 && digit(at(q)) && letter(at(q + 1)))
                h = q + 2;
            else if (alt == 3 && digit(at(q)) && letter(at(q + 1)) && letter(at(q + 2)))
                h = q + 3;
            else
                continue;
            if (!digit(at(h)))
                continue;
            const int body = h + 1, m = run(body);
            for (int k = m; k >= 0; k--) {  // [a-z0-9]* greedy, backed off until a letter follows
                if (!letter(at(body + k)))
                    continue;
                int e = body + k + 1;
                if (at(e) == '/') {  // (?:/([a-z0-9]+))?
                    const int r = run(e + 1);
                    if (r > 0) {
                        e += 1 + r;
                        if (at(e) == '/' && (at(e + 1) == 'p' || at(e + 1) == 'a' || at(e + 1) == 'm'))  // (?:/(p|a|m|mm|am))?
                            e += 2;
                    }
                }
                return e;
            }
        }
        return -1;
    };
    for (int i = 0; i < n; i++) {
        if (!space(s[i]))
            continue;
        const int p = i + 1, r = run(p);
        int e = -1;
        if (r > 0 && at(p + r) == '/')  // (?:([a-z0-9]+)/)?  - only the whole run can be followed by the slash
            e = call_from(p + r + 1);
        if (e < 0)
            e = call_from(p);
        if (e >= 0) {
            *pos = i;
            *len = e - i;
            return true;
        }
    }
    return false;
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
    // FindNext(callsignExp(), ...) without the regex machinery
    bool FindNextCallsign(bool includeTail, std::string *out)
    {
        const std::string &cur = window_[current_];
        if (searchPoint_ >= (int)cur.size())
            return false;
        const int n = (int)cur.size() - searchPoint_;
        int pos = 0, len = 0;
        if (!CallsignSearch(cur.data() + searchPoint_, n, &pos, &len))
            return false;
        const int end = pos + len;
        if (!includeTail && end >= n)
            return false;
        out->assign(cur, (size_t)(searchPoint_ + pos), (size_t)len);
        searchPoint_ += end;
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

// hamradio/callsign.Parse + Callsign.String(): PREFIX/BASE/SUFFIX/WORKING_CONDITION, upper case.  The syntax as a
// regular expression (kept: the straight-line matcher below is checked against it, tests/host/test_rx_host.cpp "cpu"):
inline const std::regex &callsignSyntax()
{
    static const std::regex re(
        R"(^(?:([A-Z0-9]+)/)?((?:[A-Z]|[A-Z][A-Z]|[0-9][A-Z]|[0-9][A-Z][A-Z])[0-9][A-Z0-9]*[A-Z])(?:/([A-Z0-9]+))?(?:/(P|A|M|MM|AM))?$)",
        std::regex::ECMAScript);
    return re;
}

// Does the whole (upper-case) string have that syntax?  An anchored match may take ANY way through the pattern, so
// every end of the call's body is tried, not only the greedy one.  Callsign.String() joins the parts it parsed with
// slashes again: for a string that matches, the canonical form is the string itself.
inline bool CallsignSyntaxOK(const std::string &up)
{
    const char *s = up.data();
    const int n = (int)up.size();
    auto letter = [](char c) { return c >= 'A' && c <= 'Z'; };
    auto digit = [](char c) { return c >= '0' && c <= '9'; };
    auto alnum = [&](char c) { return letter(c) || digit(c); };
    auto at = [&](int i) { return i < n ? s[i] : '\0'; };
    auto run = [&](int i) {
        int k = 0;
        while (i + k < n && alnum(s[i + k]))
            k++;
        return k;
    };
    // what may follow the call: nothing, "/X+", "/X+/WC" (a lone "/WC" is a "/X+")
    auto tail_ok = [&](int e) {
        if (e == n)
            return true;
        if (at(e) != '/')
            return false;
        const int r = run(e + 1);
        if (r == 0)
            return false;
        const int f = e + 1 + r;
        if (f == n)
            return true;
        if (at(f) != '/')
            return false;
        const std::string wc = up.substr((size_t)f + 1);
        return wc == "P" || wc == "A" || wc == "M" || wc == "MM" || wc == "AM";
    };
    auto call_from = [&](int q) {
        for (int alt = 0; alt < 4; alt++) {
            int h;
            if (alt == 0 && letter(at(q)))
                h = q + 1;
            else if (alt == 1 && letter(at(q)) && letter(at(q + 1)))
                h = q + 2;
            else if (alt == 2 && digit(at(q)) && letter(at(q + 1)))
                h = q + 2;
            else if (alt == 3 && digit(at(q)) && letter(at(q + 1)) && letter(at(q + 2)))
                h = q + 3;
            else
                continue;
            if (!digit(at(h)))
                continue;
            const int body = h + 1, m = run(body);
            for (int k = m; k >= 0; k--)
                if (letter(at(body + k)) && tail_ok(body + k + 1))
                    return true;
        }
        return false;
    };
    const int r = run(0);
    if (r > 0 && at(r) == '/' && call_from(r + 1))
        return true;
    return call_from(0);
}

inline bool ParseCallsign(const std::string &s, std::string *canonical)
{
    std::string up;
    for (char c : s)
        up.push_back((char)std::toupper((unsigned char)c));
    if (!CallsignSyntaxOK(up))
        return false;
    *canonical = up;
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
        if (window_.FindNextCallsign(true, &candidate))
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
            if (window_.FindNextCallsign(false, &candidate))
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
