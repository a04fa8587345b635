"""Time of one same-size 3x3 convolution layer (a one-layer program) against the number of images:
python tools/bench_conv_layer.py [channels] [height=width] [rows ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vae_conditional_sampling_amd import _lib  # noqa: E402

if os.environ.get("CSVAE_EXP_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["CSVAE_EXP_LIB"])
c = int(sys.argv[1]) if len(sys.argv) > 1 else 64
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 28
rows_list = [int(v) for v in sys.argv[3:]] or [125, 250, 500, 1000, 2000, 4000]
g = torch.Generator().manual_seed(0)
b = _lib.ProgramBuilder(c * hw * hw, c * hw * hw, 1)
# two layers, so that the second finds its input's per-image maxima written by the first (the fp16 split's steady state)
for i, (src, dst) in enumerate(((0, 1), (1, _lib.CS_NET_BUF_OUT))):
    b.conv(src, dst, torch.randn(c, c, 3, 3, generator=g) / (3 * c ** 0.5), torch.randn(c, generator=g) * 0.1,
           hin=hw, win=hw, relu=True)
net = b.build()
for rows in rows_list:
    x = torch.randn(rows, c * hw * hw, generator=g).cuda()
    net.reserve(rows)
    for _ in range(3):
        net.forward(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        net.forward(x)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n / 2
    print(f"{c} channels {hw}x{hw}, {rows} images: {us:8.1f} us per layer, {us / rows:6.3f} us per image, "
          f"{2 * 9 * c * c * hw * hw * rows / us / 1e6:7.1f} TFLOP/s")
