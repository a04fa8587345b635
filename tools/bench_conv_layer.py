"""Time of single same-size 3x3 convolution layers (the halo kernels) through a layer program of N equal layers.

    python tools/bench_conv_layer.py [rows] [layers] [channels[:resolution],...]

Shapes: the residual stages of the reference's ResNet encoders / decoders (channels x resolution).  Prints the time per
layer and the algorithmic TFLOP/s (2 * 9 * cin * cout * H * W per image) against the arithmetic's ceiling.
"""
import sys
import time

import torch

sys.path.insert(0, ".")
from vae_conditional_sampling_amd import _lib  # noqa: E402

SHAPES = [(64, 28), (128, 14), (256, 7), (512, 4), (256, 4), (128, 7), (64, 14), (32, 14)]


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    nl = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    arith = _lib.get_mlp_arith()
    peak = 157.3 if arith == "f32" else 2500.0 / 6
    g = torch.Generator().manual_seed(0)
    only = sys.argv[3].split(",") if len(sys.argv) > 3 else None
    for c, hw in SHAPES:
        if only is not None and str(c) not in only and f"{c}:{hw}" not in only:
            continue
        b = _lib.ProgramBuilder(c * hw * hw, c * hw * hw, 1)
        w = torch.randn(c, c, 3, 3, generator=g) / (3 * c ** 0.5)
        bias = torch.randn(c, generator=g) * 0.1
        src = 0
        for i in range(nl):
            dst = _lib.CS_NET_BUF_OUT if i == nl - 1 else 1 + (i & 1)
            b.conv(src, dst, w, bias, hin=hw, win=hw, relu=True)
            src = dst
        net = b.build()
        x = torch.randn(rows, c * hw * hw, generator=g).cuda()
        net.reserve(rows)
        # the shader clock needs a few hundred milliseconds of load to settle: warm up for 0.4 s, time 0.4 s
        t0 = time.perf_counter()
        n1 = 0
        while time.perf_counter() - t0 < 0.4:
            net.forward(x)
            torch.cuda.synchronize()
            n1 += 1
        n = max(5, n1)
        t0 = time.perf_counter()
        for _ in range(n):
            net.forward(x)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n / nl * 1e3
        fl = 2.0 * 9 * c * c * hw * hw * rows
        print(f"{c:4d} ch {hw:2d}x{hw:<2d} rows={rows}: {ms * 1e3:8.1f} us/layer  {fl / ms / 1e9:6.1f} TFLOP/s "
              f"({fl / ms / 1e9 / peak * 100:.1f}% of the {arith} ceiling)", flush=True)


if __name__ == "__main__":
    main()
