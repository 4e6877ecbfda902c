#!/usr/bin/env python3
"""IQ-strainer throughput benchmark (BASELINE.json metric: IQ MSamples/s through FFT + peak scan +
per-peak envelope / Morse decode, at 1/2/4/8 GPUs; % of the HBM roofline).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU.  Without a launcher (`WORLD_SIZE` unset) `--gpus N` with N > 1 starts its own N rank processes
before anything touches the GPU (spawn_ranks) and relays rank 0's line.  A *step* is one pass of the whole hot path (all seven kernels) over one batch of
synthetic IQ that is already resident in HBM.  Workload at every N: BASELINE config 3 — one 2 MS/s
band, 16384-point FFT, 256 tracked CW signals — per GPU (config 4 = 8 x config 3, one band per GPU):
bands are independent receivers, so the job shards by band with no data-path collective (weak
scaling); the only collective is the RCCL broadcast of the shared configuration / threshold struct
from rank 0 before the timed region.

Rank 0 prints ONE JSON line (contract in the task statement) carrying, besides the throughput:
  roofline      dominant kernel (k_fft_psd) algorithmic bytes (8 B per IQ sample) per launch over
                its average launch duration measured with HIP events on the launch stream, vs 8 TB/s
  cpu_baseline  the CPU oracle (port of the Go path) timed on this box's host cores on a bounded
                sample of the same workload
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_SAMPLE = 8   # one complex64 IQ sample read once
# float64 vector rate without FMA (the bit-exact contract forbids fusing): MI355X's 78.6 TFLOP/s
# FP64 vector peak counts an FMA as two, so 39.3 T lane-operations per second = 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz
F64_PEAK_TOPS = 39.3
# what the chip SUSTAINS on independent v_mul_f64 / v_add_f64 streams with every CU full (tools/ubench_f64.hip, round 5,
# gpurun_out -> profiles/r05_ubench_f64.txt): 30.8 T at four waves per SIMD, 29.1 at two, 22.4 at one - the clock comes
# down under float64 load (1.8 - 1.9 GHz), so the spec-clock figure above is not reachable by any kernel
F64_MEASURED_TOPS = 30.0

WORKLOADS = {
    # name: (sample_rate, block_size, tracked signals per band, bands per GPU, free_last_window)
    "c2": (192_000, 4096, 16, 1, False),
    "c3": (2_000_000, 16384, 256, 1, True),
    "c5": (2_000_000, 8192, 16, 8, False),
}
# frames per band per step.  A step is one batch through the whole path; its size is the caller's choice (the bank takes
# any), and the fixed costs of a batch - the ramp and the tail of each kernel's grid over the 256 CUs, the serial chains'
# latency - are spread over it: config 3 at 2048 / 4096 / 8192 frames per batch runs 152 / 156 / 160 GS/s in the steady
# state and 141 / 146 / 147 over a 20-step run, where the last batch's trip through the tail stages is in the timed region.
DEFAULT_FRAMES = {"c2": 4096, "c3": 8192, "c5": 2048}
WORKLOAD_TEXT = {
    "c2": "BASELINE config 2: one TCI-shaped IQ stream, 192 kS/s, 4096-pt FFT, 16 tracked peaks",
    "c3": "BASELINE config 3: wideband synthetic IQ, 2 MS/s, 16384-pt FFT, 256 concurrent CW peaks, one band per GPU "
          "(config 4 at N>1: N independent bands, one per GPU)",
    "c5": "BASELINE config 5: 8 channels per GPU x 8192-pt FFT x 16 peaks per channel (64 channels / 1024 peaks at 8 GPUs)",
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c3")
    ap.add_argument("--frames", type=int, default=None,
                    help="frames per band per step (batch); default per workload: " +
                         ", ".join(f"{k} {v}" for k, v in sorted(DEFAULT_FRAMES.items())))
    ap.add_argument("--ring", type=int, default=3, help="distinct input batches cycled through (defeats cache reuse)")
    ap.add_argument("--settle-ms", type=float, default=500.0,
                    help="untimed run-in before the warmup steps so the GPU's clocks have left their idle state "
                         "(a 20-step run measured from idle reads 15 %% slower than the steady state)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="target CPU work per core for the baseline")
    ap.add_argument("--kernel-breakdown", action="store_true", help="print per-kernel HIP-event times to stderr")
    ap.add_argument("--serial", action="store_true",
                    help="diagnostic: run all stages on one stream (no overlap) so per-kernel times are standalone")
    ap.add_argument("--graph", action="store_true",
                    help="A/B: the steady state as a captured hipGraph (sdr_graph_*), one replay = sdr_graph_batches() steps; "
                         "--steps and --warmup are rounded up to whole replays")
    ap.add_argument("--no-delivery", action="store_true",
                    help="diagnostic: leave results in HBM (no sdr_poll in the timed loop), as round 1 measured")
    return ap.parse_args()


def f64_ops_per_sample(n, r32=False):
    """float64 add / mul instructions per IQ sample in the FFT kernel and in the whole path, counted in the compiled
    kernels (tools/count_f64_ops.py for k_fft_psd; k_fft_r32's frame loop in its assembly: profiles/f64_ops.json);
    None if the count is missing for this block size."""
    path = os.path.join(ROOT, "profiles", "f64_ops.json")
    try:
        d = json.load(open(path))[f"{n}_r32" if r32 else str(n)]
        return d["k_fft_psd"], d["whole_path"]
    except (OSError, KeyError, ValueError):
        return None, None


def measured_traffic(workload, frames):
    """HBM bytes per k_fft_psd launch from the committed rocprofv3 --pmc runs (profiles/traffic.json), but only if
    they were measured on exactly the kernel sources this run was built from (a plain bench run cannot collect
    PMC counters itself); otherwise None."""
    from sdrainer_amd.csrc import build

    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        if tj.get("_source_hash") != build.source_hash():
            return None
        return tj[f"{workload}_f{frames}"]["k_fft_psd_hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def available_cores() -> int:
    """Host cores this process may actually use (affinity mask and cgroup quota), capped at 16 — the
    CPU share of a one-GPU box."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def cpu_baseline(rate, n, tones, free_last, seconds):
    """Times oracle (port of the Go path: complex128 radix-2 FFT + log10 projection + noise floor +
    thresholds + listeners + cumulation + FindPeaks) with one band per host core."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import oracle as orc
    from sdrainer_amd import synth

    cores = available_cores()
    block_frames = 128
    iq, bins, _ = synth.make_band(block_frames, rate, n, tones, seed=4242, free_last_window=free_last)
    edge = synth.default_edge_width(n)

    def make():
        r = orc.Receiver(rate, n, edge)
        for b in bins:
            r.attach(int(b))
        return r

    # calibrate on one core
    r = make()
    t0 = time.perf_counter()
    r.run_baseline(iq)
    dt = time.perf_counter() - t0
    one_core = block_frames * n / dt / 1e6
    reps = max(1, int(seconds / dt))

    def work(_):
        rr = make()
        for _ in range(reps):
            rr.run_baseline(iq)  # ctypes releases the GIL: threads run in parallel
        return reps * block_frames * n

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        total = sum(ex.map(work, range(cores)))
    dt = time.perf_counter() - t0
    return {
        "value": round(total / dt / 1e6, 2),
        "unit": "MSamples/s",
        "cores": cores,
        "kind": "port",
        "value_1core": round(one_core, 2),
        "sample": f"{cores} bands (one per host core) x {reps * block_frames} frames of {n} samples "
                  f"({block_frames}-frame block replayed {reps}x), {tones} listeners each, oracle/sdr_oracle.c "
                  f"(C restatement of the Go path, -O2 -ffp-contract=off); the Go binary itself cannot be built here",
    }


def spawn_ranks(n: int, deadline_s: float = 1500.0) -> int:
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes of this same script (one per GPU,
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, as torch.distributed.run would set them), BEFORE
    this process has touched the GPU - it never does.  Rank 0's stdout (the one JSON line) is relayed; the exit code
    is non-zero if any rank failed.  The ranks are watched together: when one dies (a bad LOCAL_RANK on a box with fewer
    GPUs, a failed rendezvous) the others - which would sit in init_process_group or a barrier for ever, holding their
    GPUs - are terminated, then killed, and the tail of every rank's output goes into the error message; the whole wait
    has a deadline."""
    import socket
    import subprocess
    import tempfile

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs, logs = [], []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), SDR_BENCH_SPAWNED="1")
        out = tempfile.TemporaryFile()
        err = tempfile.TemporaryFile()
        logs.append((out, err))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out, stderr=err))

    def tail(f, nbytes=1500):
        f.seek(0, os.SEEK_END)
        size = f.tell()
        f.seek(max(0, size - nbytes))
        return f.read().decode(errors="replace")

    t_end = time.monotonic() + deadline_s
    failed, timed_out = None, False
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = bad
            break
        if all(c == 0 for c in codes):
            break
        if time.monotonic() > t_end:
            timed_out = True
            break
        time.sleep(0.05)
    if failed or timed_out:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_kill = time.monotonic() + 10.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_kill - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        why = f"ranks failed (rank, exit code): {failed}" if failed else f"no result after {deadline_s:.0f} s"
        print(f"bench.py: {why}; the other ranks were stopped", file=sys.stderr)
        for r, (out, err) in enumerate(logs):
            print(f"--- rank {r} (exit {procs[r].returncode}) stderr tail:\n{tail(err)}\n--- rank {r} stdout tail:\n{tail(out)}", file=sys.stderr)
        return 1
    logs[0][0].seek(0)
    sys.stdout.write(logs[0][0].read().decode(errors="replace"))
    sys.stdout.flush()
    sys.stderr.write(tail(logs[0][1], 4000))  # (rank 0's diagnostics, e.g. --kernel-breakdown)
    return 0


def pin_to_gpu_numa_node(torch, local_rank: int, rank: int, world: int):
    """Keep this rank's threads - the enqueueing main thread, the consumer thread in sdr_poll, the library's staging
    copies - on the cores of its GPU's NUMA node (eight ranks x several threads otherwise wander over both sockets and
    poll pinned memory across the link).  Falls back to an even split of the cores when the node is not exposed.
    Returns what was done, for the result line."""
    try:
        all_cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return {"policy": "none (no sched_setaffinity)"}
    cpus, policy = None, None
    try:
        pr = torch.cuda.get_device_properties(local_rank)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        base = f"/sys/bus/pci/devices/{bdf}"
        node = int(open(base + "/numa_node").read())
        if node >= 0:
            cl = open(f"/sys/devices/system/node/node{node}/cpulist").read().strip()
            want = set()
            for part in cl.split(","):
                lo, _, hi = part.partition("-")
                want.update(range(int(lo), int(hi or lo) + 1))
            cpus = sorted(want & set(all_cpus))
            policy = f"numa node {node} of {bdf}"
    except (OSError, ValueError, AttributeError):
        pass
    if world > 1 and cpus:
        # the ranks that share the node share its cores evenly (rank order)
        share = max(1, len(cpus) // max(1, min(world, 4)))
        k = rank % max(1, len(cpus) // share)
        cpus = cpus[k * share:(k + 1) * share] or cpus
    if not cpus and world > 1:
        share = max(1, len(all_cpus) // world)
        cpus = all_cpus[rank * share:(rank + 1) * share] or all_cpus
        policy = "even split (GPU NUMA node not exposed)"
    if not cpus:
        return {"policy": "none (single rank, NUMA node not exposed)", "cpus": len(all_cpus)}
    os.sched_setaffinity(0, cpus)
    return {"policy": policy, "cpus": len(cpus), "first": cpus[0], "last": cpus[-1]}


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args.gpus))
    if args.serial:
        os.environ["SDR_NO_OVERLAP"] = "1"
    # (tests/test_bench_spawn.py: a rank that dies / a rank that hangs in front of the rendezvous, without a GPU)
    if os.environ.get("SDR_BENCH_TEST_DIE_RANK") == os.environ.get("RANK", "0"):
        raise SystemExit(3)
    if os.environ.get("SDR_BENCH_TEST_HANG_RANK") == os.environ.get("RANK", "0"):
        time.sleep(3600)
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:  # never report an n_gpus that is not what was asked for
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # rehearsal knobs (a one-GPU box cannot host two RCCL ranks): SDR_DIST_BACKEND=gloo runs the
    # collectives on the CPU, SDR_FORCE_DEVICE pins every rank to one GPU
    backend = os.environ.get("SDR_DIST_BACKEND", "nccl")
    if "SDR_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["SDR_FORCE_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    affinity = pin_to_gpu_numa_node(torch, local_rank, rank, world)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")
    dist = None
    if world > 1:
        import torch.distributed as dist  # noqa: F811

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        assert dist.get_world_size() == world

    from sdrainer_amd import capi, sharding, synth

    rate, n, tones, bands_per_gpu, free_last = WORKLOADS[args.workload]
    frames = args.frames or DEFAULT_FRAMES[args.workload]
    edge = synth.default_edge_width(n)

    # the one genuinely shared piece of state: configuration / thresholds, broadcast from rank 0
    shared = sharding.SharedConfig(sample_rate=rate, block_size=n, edge_width=edge, peak_threshold=15.0,
                                   signal_debounce=1, max_listeners=max(tones, 1))
    shared = sharding.broadcast_config(shared, dist, coll_dev)
    my_bands = sharding.bands_of_rank(bands_per_gpu * world, world, rank)
    assert len(my_bands) == bands_per_gpu

    bank = capi.Bank(shared.sample_rate, shared.block_size, n_bands=bands_per_gpu, edge_width=shared.edge_width,
                     peak_threshold=shared.peak_threshold, signal_debounce=shared.signal_debounce,
                     max_listeners=shared.max_listeners, max_batch_frames=frames, max_peaks=1024, find_peaks=True,
                     trace=False, device_id=local_rank)
    # the bank behind the job-wide setters (sharding.ShardedBank: rank 0's values, applied on every rank at the same batch
    # boundary) - one collective setter call before the timed region, as a host would make when the operator turns a knob
    sbank = sharding.ShardedBank(bank, shared, dist, coll_dev, n_bands_total=bands_per_gpu * world)
    shared = sbank.set_peak_threshold(shared.peak_threshold)
    stream = torch.cuda.current_stream()
    if args.graph:
        stream = torch.cuda.Stream()  # the null stream cannot be captured
        torch.cuda.set_stream(stream)
    bank.set_stream(stream.cuda_stream)

    # inputs: `ring` distinct batches, layout [band][frame][N][2] float32, generated in HBM
    ring = []
    bins_of_band = {}
    for k in range(args.ring):
        per_band = []
        for bi, band in enumerate(my_bands):
            iq, bins, _ = synth.make_band_torch(frames, rate, n, tones, seed=1000 * 3 + 17 * band + k, device=dev,
                                                free_last_window=free_last)
            bins_of_band[bi] = bins
            per_band.append(iq)
        ring.append(torch.stack(per_band).contiguous())
    for bi in range(bands_per_gpu):
        for b in bins_of_band[bi]:
            bank.attach(bi, int(b))
    torch.cuda.synchronize()

    # results are DELIVERED inside the timed region (peaks, edges, runes of every batch into host buffers, by the
    # consumer thread below); the timed region ends when the last batch has been delivered; nothing may be dropped
    delivery = not args.no_delivery
    got = {"batches": 0, "peaks": 0, "edges": 0, "runes": 0, "runes_dropped": 0, "edges_dropped": 0}
    if delivery:
        bank.enable_results(True)

    # The consumer is a thread of its own, as the reference's Reporter runs on goroutines of its own: it blocks in
    # sdr_poll (the C call releases the GIL) and copies every finished batch's records into host buffers while the
    # main thread keeps enqueueing.
    import threading

    consumer_stop = threading.Event()

    def consume():
        while True:
            r = bank.poll_counts(wait=True)
            if r is None:
                if consumer_stop.is_set():
                    return
                time.sleep(20e-6)
                continue
            got["batches"] += 1
            got["peaks"] += r[2]
            got["edges"] += r[4]
            got["runes"] += r[5]
            got["runes_dropped"], got["edges_dropped"] = r[6], r[7]

    consumer = None
    if delivery:
        consumer = threading.Thread(target=consume, daemon=True)
        consumer.start()

    def take(wait=False):
        """wait: until everything enqueued so far has been delivered AND counted by the consumer thread (the library
        reports a batch delivered inside sdr_poll, a moment before the consumer adds it to `got`)."""
        while wait and (bank.results_pending > 0 or got["batches"] < enqueued[0]):
            time.sleep(0)  # (yields the GIL to the consumer thread; a timed sleep here rounds up to 60 - 100 us of harness time inside the timed region)

    enqueued = [0]  # batches handed to the bank so far

    def step(i):
        bank.process_device(ring[i % len(ring)].data_ptr(), frames)
        enqueued[0] += 1

    if args.graph:
        K = bank.graph_batches
        args.steps = -(-args.steps // K) * K
        args.warmup = -(-args.warmup // K) * K
        torch.cuda.synchronize()
        bank.graph_capture(frames)

        def step(i):  # noqa: F811  (one replay per K steps)
            if i % K == 0:
                bank.graph_launch([ring[(i + k) % len(ring)].data_ptr() for k in range(K)])
                enqueued[0] += K

    # run-in (untimed, not counted as warmup): the clocks ramp up from idle over the first few hundred
    # milliseconds of load; then the W warmup steps of the contract
    t_settle = time.perf_counter()
    k = 0
    while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
        for _ in range(16):
            step(k)
            k += 1
        torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if delivery:
        take(wait=True)
    base = dict(got)  # the timed region counts from here
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    if os.environ.get("SDR_BENCH_PROFILE_TIMED"):  # (development: the library's stage events over the timed region itself)
        bank.sync()
        bank.profile_enable(True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    t_enq = time.perf_counter()
    torch.cuda.synchronize()
    t_gpu = time.perf_counter()
    if delivery:
        take(wait=True)  # the last batches' results, still inside the timed region
    own_elapsed = time.perf_counter() - t0  # this rank alone: a straggling GPU shows here, not in the max below
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    delivered = {k_: got[k_] - (base[k_] if k_ in ("batches", "peaks", "edges", "runes") else 0) for k_ in got}
    if delivery:
        assert delivered["batches"] == args.steps, f"delivered {delivered['batches']} of {args.steps} batches"
        assert delivered["runes_dropped"] == 0 and delivered["edges_dropped"] == 0, delivered
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    per_rank_s = [own_elapsed]
    if dist is not None:
        own = torch.tensor([own_elapsed], dtype=torch.float64, device=coll_dev)
        every = [torch.zeros_like(own) for _ in range(world)]
        dist.all_gather(every, own)
        per_rank_s = [float(x.item()) for x in every]

    samples_per_step_rank = frames * n * bands_per_gpu
    total_samples = samples_per_step_rank * args.steps * world
    value = total_samples / elapsed / 1e6

    # dominant kernel: HIP events on the launch stream, live, same workload, after the timed region
    # (same pipelined run, continued: a run-in so the pipeline is full again, then up to 200 measured steps;
    # the first profiled steps of a drained pipeline see no co-running stages and read 10 % short)
    if args.graph:
        bank.sync()
        if delivery:
            take(wait=True)
        bank.graph_release()  # the per-kernel events below are taken on the eager path

        def step(i):  # noqa: F811
            bank.process_device(ring[i % len(ring)].data_ptr(), frames)
    bank.profile_enable(True)
    for i in range(min(args.steps, 20)):
        step(i)
    bank.sync()
    bank.profile_reset()
    prof_steps = max(3, min(args.steps, 200))
    for i in range(prof_steps):
        step(i)
    bank.sync()
    if delivery:
        take(wait=True)
    prof = bank.profile_read()
    # the same kernel with the machine to itself: one step at a time, drained in between.  (Pipelined, its launches
    # run back to back and share the CUs with the other stages, so its duration above is the step's.)
    bank.profile_reset()
    alone_steps = max(3, min(args.steps, 30))
    for i in range(alone_steps):
        step(i)
        bank.sync()
    if delivery:
        take(wait=True)
    prof_alone = bank.profile_read()
    bank.profile_enable(False)
    if consumer is not None:
        consumer_stop.set()
        consumer.join(timeout=10)
    # (the stage is called k_fft_psd in the library's profile whichever kernel serves it: N = 16384 with at most 512 listener
    # slots in use and at least 1024 frames per launch runs k_fft_r32 - 512 threads x 32 points, the next frame prefetched
    # into registers - unless SDR_FFT_R32=0: k_fft_psd.hip launch_fft)
    r32_env = os.environ.get("SDR_FFT_R32")
    r32 = n == 16384 and tones <= 512 and (r32_env not in (None, "0") or (r32_env is None and frames * bands_per_gpu >= 1024))
    fft_kernel_name = "k_fft_r32" if r32 else "k_fft_psd"
    fft_ms, fft_n = prof["k_fft_psd"]
    fft_avg_ms = fft_ms / max(fft_n, 1)
    fft_alone_ms = prof_alone["k_fft_psd"][0] / max(prof_alone["k_fft_psd"][1], 1)
    achieved = BYTES_PER_SAMPLE * samples_per_step_rank / (fft_avg_ms * 1e-3) / 1e9
    traffic = measured_traffic(args.workload, frames)
    ops_fft, ops_path = f64_ops_per_sample(n, r32)
    if args.kernel_breakdown and rank == 0:
        print("  timed region: enqueued after %.3f ms, device idle after %.3f ms, last batch delivered after %.3f ms" %
              ((t_enq - t0) * 1e3, (t_gpu - t0) * 1e3, own_elapsed * 1e3), file=sys.stderr)
        tot = sum(v[0] for v in prof.values())
        for k, (ms, cnt) in prof.items():
            print(f"  {k:16s} {ms / max(cnt, 1):9.3f} ms/launch  {100 * ms / max(tot, 1e-9):5.1f}%", file=sys.stderr)
        print(f"  sum of kernels   {tot / prof_steps:9.3f} ms/step ; wall {1e3 * elapsed / args.steps:9.3f} ms/step",
              file=sys.stderr)

    # sanity: the timed path really decoded something (guards against measuring an empty pipeline)
    if delivery:
        decoded = delivered["runes"]
        assert decoded > 0 and delivered["edges"] > 0, delivered
    else:
        decoded = sum(len(bank.read_text(0, lid)) for lid in range(min(tones, 4)))
    chunks = bank.last_batch_chunks
    # every band's record on every rank, once, after the timed region: each band of the job must have decoded something
    last_rec = bank.read_frame_records(0)[-1]
    local_records = np.stack([sharding.make_record(b, int(bank.total_frames), int(bank.total_frames) * n, chunks, delivered["edges"] if delivery else 0,
                                                   decoded, float(last_rec["noise_floor"]), float(last_rec["listen_thr"])) for b in my_bands])
    job_records = sbank.gather(local_records)
    assert [int(r[0]) for r in job_records] == list(range(bands_per_gpu * world)), job_records[:, 0]
    assert all(r[5] > 0 for r in job_records), "a band of the job decoded nothing"
    # rehearsal / test hook: what THIS rank worked on (tests/test_bench_ranks.py)
    if os.environ.get("SDR_BENCH_RANK_REPORT"):
        with open(os.environ["SDR_BENCH_RANK_REPORT"], "w") as fh:
            json.dump({"rank": rank, "bands": my_bands, "shared_config": sharding.describe(shared),
                       "decoded_runes": decoded, "job_bands": [int(r[0]) for r in job_records]}, fh)

    result = {
        "metric": "IQ MSamples/s through FFT+peak+envelope",
        "value": round(value, 1),
        "unit": "MSamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": WORKLOAD_TEXT[args.workload],
            "sample_rate": rate, "block_size": n, "tracked_signals_per_band": tones,
            "bands_per_gpu": bands_per_gpu, "frames_per_step_per_band": frames,
            "samples_per_step_per_gpu": samples_per_step_rank, "input": "complex64 IQ resident in HBM",
            "sharding": f"{bands_per_gpu * world} independent bands, {bands_per_gpu} per GPU, no data-path collective",
            "distributed": ({"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                             "collectives": "broadcast of the shared configuration struct from rank 0 before the timed "
                                            "region; barrier + max-over-ranks of the elapsed time around it; all-gather "
                                            "of every rank's own elapsed time (per_rank)",
                             "launched_by": "bench.py itself" if os.environ.get("SDR_BENCH_SPAWNED") else "external launcher"}
                            if dist is not None else {"world_size": 1, "backend": None}),
            # every rank's OWN time from the common start to its last delivered batch (the closing barrier excluded):
            # `value` is computed from the max over ranks, so a straggling GPU would otherwise be invisible
            "per_rank": {"ms_per_step": [round(t_ * 1e3 / args.steps, 4) for t_ in per_rank_s],
                         "value": [round(samples_per_step_rank * args.steps / t_ / 1e6, 1) for t_ in per_rank_s],
                         "value_min": round(samples_per_step_rank * args.steps / max(per_rank_s) / 1e6, 1),
                         "value_max": round(samples_per_step_rank * args.steps / min(per_rank_s) / 1e6, 1)},
            "cpu_affinity_rank0": affinity,
            "clock_settle_ms": args.settle_ms,
            "launch": (f"hipGraph: {bank.graph_batches} batches per replay (sdr_graph_launch)" if args.graph
                       else "eager: every kernel launched per step over the bank's four streams"),
            "delivery": ("inside the timed region: a consumer thread blocks in sdr_poll and copies every batch's peaks, "
                         "keying edges and decoded runes to host buffers; the timer stops when the last batch has "
                         "been delivered; drop counters asserted zero") if delivery
                        else "none (results left in HBM)",
            "sanity": ({"batches_delivered": delivered["batches"], "peaks": delivered["peaks"], "edges": delivered["edges"],
                        "runes": delivered["runes"], "cumulations_per_step": chunks} if delivery
                       else {"runes_decoded_first_listeners": decoded, "cumulations_per_step": chunks}),
        },
        "roofline": {
            "bound": "hbm", "kernel": fft_kernel_name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "avg_launch_ms": round(fft_avg_ms, 4), "launches_timed": fft_n,
            "standalone": {"avg_launch_ms": round(fft_alone_ms, 4), "launches_timed": prof_alone["k_fft_psd"][1],
                           "frac": round(BYTES_PER_SAMPLE * samples_per_step_rank / (fft_alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            "whole_path_frac": round(value * 1e6 / world * BYTES_PER_SAMPLE / (HBM_PEAK_GBS * 1e9), 4),
        },
    }
    if ops_fft:
        # What actually bounds the kernel: float64 vector issue.  The bit-exact contract (same butterfly graph, no
        # FMA) fixes the float64 operations per sample, so the chip's non-FMA float64 rate caps the sample rate,
        # and with it the reachable fraction of the HBM roofline (ceiling_hbm_frac), below the 8 B/sample line.
        tops = ops_fft * samples_per_step_rank / (fft_avg_ms * 1e-3) / 1e12
        result["roofline"]["compute"] = {
            "bound": "f64_valu", "ops_per_sample": ops_fft, "achieved_Tops": round(tops, 2), "peak_Tops": F64_PEAK_TOPS,
            # (the spec clock; inside k_fft_psd the shader clock reads 2.25 GHz - profiles/*_fft_standalone.txt - where the
            # same count gives 36.9: fractions of the peak below are about 6 % higher against that)
            "peak_Tops_at_measured_clock": round(F64_PEAK_TOPS * 2.25 / 2.4, 1),
            "measured_peak_Tops": F64_MEASURED_TOPS,
            "frac": round(tops / F64_PEAK_TOPS, 4),
            "frac_of_measured_peak": round(tops / F64_MEASURED_TOPS, 4),
            "ceiling_hbm_frac_at_measured_peak": round(F64_MEASURED_TOPS * 1e12 / ops_fft * BYTES_PER_SAMPLE / (HBM_PEAK_GBS * 1e9), 4),
            "ceiling_hbm_frac": round(F64_PEAK_TOPS * 1e12 / ops_fft * BYTES_PER_SAMPLE / (HBM_PEAK_GBS * 1e9), 4),
            "whole_path_ops_per_sample": ops_path,
            "whole_path_ceiling_hbm_frac": round(F64_PEAK_TOPS * 1e12 / ops_path * BYTES_PER_SAMPLE / (HBM_PEAK_GBS * 1e9), 4),
            "whole_path_frac_of_ceiling": round(value * 1e6 / world * ops_path / (F64_PEAK_TOPS * 1e12), 4),
        }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(rate, n, tones, free_last, args.cpu_seconds)
    bank.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
