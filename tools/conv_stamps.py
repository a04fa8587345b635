"""Where a workgroup of the halo convolution kernel spends its time (debug build with -DCS_CONV_STAMPS only).

    CSVAE_EXP_LIB=exp/libcsvae_stamps.so python tools/conv_stamps.py [rows] [channels:resolution]

Runs a layer program of equal same-size 3x3 convolutions, reads the stamps that wave 0 of every workgroup of the last
launch left (100 MHz clock: entry, window staged, last matrix instruction issued, stores issued; shader clock over the
loop) and prints the phase durations and how the workgroups are spread over the launch.
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from vae_conditional_sampling_amd import _lib  # noqa: E402

if os.environ.get("CSVAE_EXP_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["CSVAE_EXP_LIB"])


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    c, hw = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "64:28").split(":"))
    res = len(sys.argv) > 3 and sys.argv[3] == "res"
    g = torch.Generator().manual_seed(0)
    b = _lib.ProgramBuilder(c * hw * hw, c * hw * hw, 1)
    w = torch.randn(c, c, 3, 3, generator=g) / (3 * c ** 0.5)
    bias = torch.randn(c, generator=g) * 0.1
    nl, src = 4, 0
    for i in range(nl):
        dst = _lib.CS_NET_BUF_OUT if i == nl - 1 else 1 + (i & 1)
        b.conv(src, dst, w, bias, hin=hw, win=hw, relu=True)
        src = dst
    net = b.build()
    x = torch.randn(rows, c * hw * hw, generator=g).cuda()
    net.reserve(rows)
    for _ in range(40):
        net.forward(x)
    torch.cuda.synchronize()
    L = _lib.lib()
    px = int(os.environ.get("CS_STAMP_PX", 512 if c == 64 else 256))
    nb = min(8192, (rows * hw * hw + px - 1) // px)
    buf = (C.c_ulonglong * (8 * nb))()
    L.cs_debug_conv_stamps.argtypes = [C.c_void_p, C.c_int]
    rc = L.cs_debug_conv_stamps(buf, 8 * nb)
    assert rc == 0, rc
    s = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 8).astype(np.int64)
    t0 = s[:, 0].min()
    us = (s[:, :4] - t0) / 100.0
    print(f"{c} ch {hw}x{hw}, {rows} images: {nb} workgroups; launch span {us[:, 3].max():.1f} us")
    pro, loop, epi = us[:, 1] - us[:, 0], us[:, 2] - us[:, 1], us[:, 3] - us[:, 2]
    cyc = (s[:, 5] - s[:, 4]).astype(np.float64)
    for name, v in (("prologue", pro), ("loop", loop), ("epilogue", epi), ("whole workgroup", us[:, 3] - us[:, 0])):
        print(f"  {name:16s} mean {v.mean():7.2f} us   p10 {np.percentile(v, 10):7.2f}   p90 {np.percentile(v, 90):7.2f}")
    print(f"  loop in shader cycles: mean {cyc.mean():.0f}  -> clock {cyc.mean() / loop.mean() / 1e3:.2f} GHz")
    # the workgroups in order of their entry: rounds of 256
    order = np.argsort(us[:, 0])
    for r in range(0, min(nb, 256 * 4), 256):
        sel = order[r:r + 256]
        print(f"  entries {r:5d}..{r + len(sel) - 1:5d}: entry {us[sel, 0].min():7.1f}..{us[sel, 0].max():7.1f} us,"
              f" exit {us[sel, 3].min():7.1f}..{us[sel, 3].max():7.1f} us")


if __name__ == "__main__":
    main()
