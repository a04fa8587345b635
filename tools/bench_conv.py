"""Throughput of the ResNet encoder / decoder layer programs at the reference's full sizes.

    python tools/bench_conv.py [omniglot|mnist_gmm] [rows] [enc|dec]

FLOPs per image are the algorithmic counts (2*k*k*cin*cout*hout*wout per convolution, 2*in*out per
linear layer) summed over the program's layers.
"""
import sys
import time

import torch

sys.path.insert(0, ".")
from vae_conditional_sampling_amd import _lib  # noqa: E402
from vae_conditional_sampling_amd.models.vae_resnet import ResNetDecoder, ResNetEncoder  # noqa: E402

NETS = {
    "omniglot": dict(enc=dict(input_shape=(1, 28, 28), layers=[2, 2, 2, 2], layer_widths=[64, 128, 256, 512], latent_dim=100),
                     dec=dict(output_shape=(1, 28, 28), layers=[2, 2, 2, 2], layer_upscale=[7, None, None, None],
                              layer_widths=[256, 128, 64, 64], downscaled_resolution=(4, 4), latent_dim=50, upscale_factor=7)),
    "mnist_gmm": dict(enc=dict(input_shape=(1, 14, 14), layers=[2, 2, 2, 2], layer_widths=[32, 64, 128, 256], latent_dim=50),
                      dec=dict(output_shape=(2, 14, 14), layers=[2, 2, 2, 2], layer_upscale=[4, 7, None, None],
                               layer_widths=[128, 64, 32, 32], downscaled_resolution=(1, 1), latent_dim=25, upscale_factor=4)),
}


def program_flops(net):
    f = 0
    for L in net._program().layers:
        if L.op in (_lib.CS_OP_CONV, _lib.CS_OP_LINEAR):
            f += 2 * L.ksize * L.ksize * L.cin * L.cout * L.hout * L.wout
    return f


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "omniglot"
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    torch.manual_seed(0)
    enc = ResNetEncoder(**NETS[name]["enc"]).cuda().eval()
    dec = ResNetDecoder(**NETS[name]["dec"]).cuda().eval()
    x = torch.randn(rows, enc.input_dim, device="cuda")
    z = torch.randn(rows, dec.input_dim, device="cuda")
    which = sys.argv[3] if len(sys.argv) > 3 else None
    for label, net, inp in (("enc", enc, x), ("dec", dec, z)):
        if which is not None and label != which:
            continue
        fl = program_flops(net)
        # the shader clock needs a few hundred milliseconds of load to settle: warm up for 0.5 s
        t0 = time.perf_counter()
        n1 = 0
        while time.perf_counter() - t0 < 0.5:
            net(inp)
            torch.cuda.synchronize()
            n1 += 1
        n = max(5, n1)
        t0 = time.perf_counter()
        for _ in range(n):
            net(inp)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        from vae_conditional_sampling_amd import _lib
        arith = _lib.get_mlp_arith()
        # algorithmic TFLOP/s ceilings: exact f32; dense bf16 peak / 6 term products (3-term bf16 split); under the
        # fp16 split the 3x3 halo layers run 3 term products (dense fp16 peak / 3), the other layers keep the bf16 split
        peak = 157.3 if arith == "f32" else 2500.0 / 6
        half = 2500.0 / 3 * 1e12 if arith == "f16x2" else None
        ideal = _lib.layerwise_ideal_seconds(net._program().layers, rows, peak * 1e12, fp16_split_flops_per_s=half) * 1e3
        mix = "" if half is None else " (fp16-split halo layers at 833.3)"
        print(f"{name} {label}: rows={rows} {fl / 1e6:.1f} MFLOP/img  {ms:.3f} ms  {fl * rows / ms / 1e9:.1f} algorithmic TFLOP/s "
              f"({fl * rows / ms / 1e9 / peak * 100:.1f}% of the 3-term bf16 split ceiling {peak:.1f}; arithmetic {arith}); layer-wise "
              f"roofline (each layer at the larger of its matrix time{mix} and its HBM time at 8 TB/s) {ideal:.3f} ms = {ideal / ms * 100:.1f}%")


if __name__ == "__main__":
    main()
