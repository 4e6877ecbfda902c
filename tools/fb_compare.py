"""Development aid: two psd dumps of tools/fft_bench (SDR_FB_DUMP) compared word by word: how many float32 psd values
differ, by how many ulps.  usage: python tools/fb_compare.py a.bin b.bin"""
import sys

import numpy as np

a = np.fromfile(sys.argv[1], np.int32).astype(np.int64)
b = np.fromfile(sys.argv[2], np.int32).astype(np.int64)
d = np.abs(a - b)
print(f"{a.size} psd words, {np.count_nonzero(d)} differ ({np.count_nonzero(d) / a.size:.3e}), max {d.max()} ulp, "
      f"by 1 ulp {np.count_nonzero(d == 1)}, by more {np.count_nonzero(d > 1)}")
