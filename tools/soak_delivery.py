#!/usr/bin/env python3
"""Development aid: a long run of the delivery machinery under an erratic consumer - thousands of batches, eager and as
graph replays, the consumer polling in fits and starts so that the producer keeps finding unpolled batches in the sets it
wants back (yield-or-park, host/delivery.h Delivery::park).  No oracle at this length: the two modes must deliver the same
stream (a running hash over every batch's edges, runes and peaks, in order) and nothing may be dropped or reordered.
  python tools/soak_delivery.py [replays]"""
import hashlib
import os
import random
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(mode, replays, dev, bins, rate, n, per, tones, edge):
    import torch

    from sdrainer_amd import capi

    bank = capi.Bank(rate, n, edge_width=edge, max_batch_frames=per, max_listeners=tones, max_peaks=128)
    stream = torch.cuda.Stream()
    bank.set_stream(stream.cuda_stream)
    for b in bins:
        bank.attach(0, int(b))
    bank.enable_results(True)
    K = bank.graph_batches
    if mode == "graph":
        bank.graph_capture(per)
    nb = dev.shape[0] // per
    h, got, stop, err = hashlib.sha256(), [0], threading.Event(), []
    rng = random.Random(7)

    def consume():
        try:
            while True:
                res = bank.poll(wait=rng.random() < 0.5, copy=False)
                if res is None:
                    if stop.is_set() and bank.results_pending == 0:
                        return
                    time.sleep(rng.choice([0.0, 1e-4, 1e-3]))
                    continue
                assert res["batch_index"] == got[0] and res["first_frame"] == got[0] * per, (res["batch_index"], got[0])
                assert res["runes_dropped"] == 0 and res["edges_dropped"] == 0
                for k in ("chunks", "peaks", "listeners", "edges", "runes", "rune_frames"):
                    h.update(res[k].tobytes())
                got[0] += 1
                if rng.random() < 0.05:
                    time.sleep(rng.choice([1e-3, 4e-3]))
        except Exception as e:  # noqa: BLE001
            err.append(e)

    t = threading.Thread(target=consume)
    t.start()
    t0 = time.perf_counter()
    for rep in range(replays):
        ptrs = [dev[((rep * K + k) % nb) * per].data_ptr() for k in range(K)]
        if mode == "graph":
            bank.graph_launch(ptrs)
        else:
            for p in ptrs:
                bank.process_device(p, per)
    bank.sync()
    stop.set()
    t.join(timeout=120)
    dt = time.perf_counter() - t0
    assert not t.is_alive() and not err, err
    assert got[0] == replays * K, (got[0], replays * K)
    if mode == "graph":
        bank.graph_release()
    bank.close()
    return h.hexdigest(), dt


def main():
    import torch

    from sdrainer_amd import capi, synth

    capi.load()
    replays = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    rate, n, per, tones = 96000, 1024, 70, 3
    iq, bins, _ = synth.make_band(66 * per, rate, n, tones, seed=99)
    dev = torch.from_numpy(iq).cuda()
    edge = synth.default_edge_width(n)
    out = {}
    for mode in ("eager", "graph"):
        out[mode] = run(mode, replays, dev, bins, rate, n, per, tones, edge)
        print(f"{mode}: {replays * 6} batches in {out[mode][1]:.2f} s, stream hash {out[mode][0][:16]}", flush=True)
    assert out["eager"][0] == out["graph"][0], "the two modes delivered different streams"
    print("ok")


if __name__ == "__main__":
    main()
