"""Ratio and speed of the row-stride deflate encoder against zlib on imputation-like data, plus a decode check.
python tools/deflate_rows_check.py"""
import ctypes as C
import os
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vae_conditional_sampling_amd.utils import artifacts  # noqa: E402

rng = np.random.default_rng(0)
T, B, K, D = 256, 500, 4, 43
Xt = rng.standard_normal((1, B, 1, D)).astype(np.float32)
M = rng.random((1, B, 1, D)) < 0.5
a = np.where(M, Xt, rng.standard_normal((T, B, K, D)).astype(np.float32)).astype(np.float32)
raw = a.tobytes()
t = time.perf_counter()
c = artifacts.deflate_rows(memoryview(raw), D * 4, 0) + zlib.compressobj(6, zlib.DEFLATED, -15).flush(zlib.Z_FINISH)
dt = time.perf_counter() - t
assert zlib.decompress(c, -15) == raw
print(f"row-stride encoder: ratio {len(c) / len(raw):.4f}, {len(raw) / dt / 1e6:.0f} MB/s on one thread")
t = time.perf_counter()
z = zlib.compress(raw[:32 << 20], 6)
dt = time.perf_counter() - t
print(f"zlib level 6:       ratio {len(z) / (32 << 20):.4f}, {(32 << 20) / dt / 1e6:.0f} MB/s on one thread")
