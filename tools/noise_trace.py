"""Development aid: consumer-side timing of one k_noise_stats workgroup (diagnostic library libnoisetr.so)."""
import ctypes, os, subprocess, sys
os.environ["SDR_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "abl", "libnoisetr.so")
sys.argv = ["bench.py", "--no-cpu-baseline", "--steps", "50", "--warmup", "10", "--serial", "--settle-ms", "200"]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
bench.main()
lib = ctypes.CDLL(os.environ["SDR_HIP_LIB"])
out = (ctypes.c_ulonglong * 8)()
print("rc", lib.sdr_debug_noise_trace(out))
total, wait, tiles, spins = out[0], out[1], out[2], out[3]
print(f"consumer of workgroup 7: {total/100:.1f} us total, {wait/100:.1f} us waiting for tiles, {tiles} tiles, {spins} spins; "
      f"{(total-wait)/100/max(tiles,1)*1000:.0f} ns per tile busy, {total/100/max(tiles,1)*1000:.0f} ns per tile overall")
