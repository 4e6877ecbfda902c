"""CPU restatement of rx.TextProcessor (rx/text_processor.go) — TEST INFRASTRUCTURE ONLY.

Only tests/ may import this module; the product's host mirror is sdrainer_amd/csrc/host/text_processor.h.

Pinned by the reference's own tests (rx/text_processor_test.go), replayed in
tests/test_text_processor.py: textWindow Write/Shift/FindNext/IncludeTail tables, CollectCallsign
("cq cq cq de dl1abc dl1abc dl1abc pse k" -> DL1ABC x3) and WriteTimeout.

callsign.Parse / Callsign.String() belong to github.com/ftl/hamradio v0.2.9 (go.mod; not vendored under
/root/reference).  Its published syntax is restated: optional "PREFIX/", base call
(1-2 letters or digit+1-2 letters, a digit, then alphanumerics ending in a letter), optional "/SUFFIX",
optional working condition /P /A /M /MM /AM; String() joins the parts upper-cased.  With the DXCC and
Supercheck finders absent (the reference's nil-finder branches, rx/text_processor.go:297-303,321-324)
every parsed call is valid and has weight 0.
"""
import re

DEFAULT_TEXT_WINDOW_SIZE = 20  # rx/text_processor.go:17
SPOTTING_THRESHOLD = 3         # :18
DEFAULT_WRITE_TIMEOUT = 5.0    # :20

# :23-25 (Go RE2 leftmost-first == Python's backtracking order for this expression: no ambiguity
# between engines arises because every alternative is tried in written order and quantifiers are greedy)
CALLSIGN_EXP = re.compile(
    r"\s(?:([a-z0-9]+)/)?(([a-z]|[a-z][a-z]|[0-9][a-z]|[0-9][a-z][a-z])[0-9][a-z0-9]*[a-z])(?:/([a-z0-9]+))?(?:/(p|a|m|mm|am))?",
    re.ASCII)
_PARSE_EXP = re.compile(
    r"^(?:([A-Z0-9]+)/)?((?:[A-Z]|[A-Z][A-Z]|[0-9][A-Z]|[0-9][A-Z][A-Z])[0-9][A-Z0-9]*[A-Z])(?:/([A-Z0-9]+))?(?:/(P|A|M|MM|AM))?$",
    re.ASCII)


def parse_callsign(s):
    m = _PARSE_EXP.match(s.upper())
    if not m:
        return None
    out = ""
    if m.group(1):
        out += m.group(1) + "/"
    out += m.group(2)
    if m.group(3):
        out += "/" + m.group(3)
    if m.group(4):
        out += "/" + m.group(4)
    return out


class TextWindow:  # :326-415
    def __init__(self, window_size):
        self.window = [b"", b""]
        self.window_size = window_size
        self.current = 0
        self.search_point = 0

    def string(self):
        return self.window[self.current].decode("latin-1")

    def reset(self):  # :348-353
        self.window = [b"", b""]
        self.current = 0

    def write(self, data):  # :355-363; returns (n, error)
        append_len = min(len(data), self.window_size - len(self.window[self.current]))
        if len(data) > 0 and append_len == 0:
            return 0, True
        self.window[self.current] += data[:append_len]
        return append_len, False

    def shift(self):  # :365-377
        other = (self.current + 1) % 2
        half = self.window_size // 2
        cur = self.window[self.current]
        start = max(0, len(cur) - half)
        append_len = min(half, len(cur) - start)
        self.window[other] = cur[start:start + append_len] if append_len > 0 else b""
        self.current = other
        self.search_point = max(0, self.search_point - start)

    def is_full(self):  # :379-381
        return len(self.window[self.current]) == self.window_size

    def find_next(self, exp, include_tail):  # :383-401
        cur = self.window[self.current]
        if self.search_point >= len(cur):
            return None
        text = cur[self.search_point:].decode("latin-1")
        m = exp.search(text)
        if m is None:
            return None
        if not include_tail and m.end() >= len(text):
            return None
        self.search_point += m.end()
        return m.group(0)


class TextProcessor:  # :57-324
    def __init__(self, now=lambda: 0.0, dxcc=None, scp=None):
        self.now = now
        self.dxcc, self.scp = dxcc, scp
        self.events = []  # ("decoded", call, count, weight) | ("spotted", call) | ("timeout", call)
        self.last_write = now()
        self.last_best = None
        self.window = TextWindow(DEFAULT_TEXT_WINDOW_SIZE)
        self.collected = {}  # call -> [weight, count]

    def restart(self):  # :175-182
        self.last_write = self.now()
        self.last_best = None
        self.window.reset()
        self.collected.clear()

    def check_write_timeout(self):  # :188-193
        if self.now() - self.last_write > DEFAULT_WRITE_TIMEOUT:
            self.write_timeout()

    def write_timeout(self):  # :195-200
        c = self.window.find_next(CALLSIGN_EXP, True)
        if c is not None:
            self._collect(c)

    def write(self, data):  # :202-242
        self.last_write = self.now()
        rest = bytes(data)
        while len(rest) > 0:
            n, err = self.window.write(rest)
            assert not err
            c = self.window.find_next(CALLSIGN_EXP, False)
            if c is not None:
                self._collect(c)
            if n <= len(rest):
                rest = rest[n:]
            if self.window.is_full():
                self.window.shift()

    def _collect(self, candidate):  # :244-280
        candidate = candidate.strip().lower()
        if candidate.startswith("tu5nn"):  # :282-295
            return
        call = parse_callsign(candidate)
        if call is None:
            return
        if self.dxcc is not None and not self.dxcc(call):
            return
        if call not in self.collected:
            self.collected[call] = [1 if (self.scp is not None and self.scp(call)) else 0, 0]
        self.collected[call][1] += 1
        w, c = self.collected[call]
        self.events.append(("decoded", call, c, w))
        best, max_count = None, SPOTTING_THRESHOLD - 1  # :305-319 (ties: key order, the Go map order is unspecified)
        for k in sorted(self.collected):
            weighted = self.collected[k][1] + self.collected[k][0]
            if max_count < weighted:
                max_count, best = weighted, k
        if best is None:
            return
        if best != self.last_best and self.last_best is not None:
            self.events.append(("timeout", self.last_best))
        self.events.append(("spotted", best))
        self.last_best = best
