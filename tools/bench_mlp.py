#!/usr/bin/env python3
"""Micro-benchmark of the fused MLP kernel (encoder / decoder launches of the LAIR step)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cs_oracle as O  # noqa: E402
from vae_conditional_sampling_amd import _lib  # noqa: E402
from vae_conditional_sampling_amd.models import VAE, ResidualFCNetwork  # noqa: E402
from vae_conditional_sampling_amd.sampling import engine  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    D, Zd, H, K, R = 43, 32, 256, 4, 1
    enc_w = O.make_resfc_weights(D, 2 * Zd, 5, H, 1)
    dec_w = O.make_resfc_weights(Zd, 2 * D, 2, H, 2)
    enc = ResidualFCNetwork(D, 2 * Zd, 5, H).load_weight_dict(enc_w)
    dec = ResidualFCNetwork(Zd, 2 * D, 2, H).load_weight_dict(dec_w)
    model = VAE(dec, "normal_with_eps", enc, "normal_with_eps").cuda()
    X = torch.randn(B, K, D, device="cuda")
    M8 = (torch.rand(B, D, device="cuda") > 0.5).to(torch.uint8)
    ws = engine.lair_workspace(model, B, K, R, 1, X.device)
    cfg = engine.lair_cfg(model, ws, weighting="dmis", scheme="gr", method="multinomial", cluster_size=0, weight_alpha=None)
    bufs = _lib.make_bufs(X=X, M=M8, q=ws.q, Z=ws.Z, gen=ws.gen, gen_ll=ws.gen_ll, log_w=ws.log_w, norm_log_w=ws.norm_log_w,
                          idx=ws.idx, ess=ws.stats[0], tess=ws.stats[1], perplexity=ws.stats[2])
    L = _lib.lib()
    he, hd = model.var_latent_network.packed().handle, model.generator_network.packed().handle
    st = _lib.current_stream()
    rng = _lib.make_rng(1, 0, 0)
    fe = 2 * (D * H + 10 * H * H + H * 2 * Zd) * B * K
    fd = 2 * (Zd * H + 4 * H * H + H * 2 * D) * B * (K + R)
    for name, fn, fl in (("enc", lambda: L.cs_lair_encode(he, C.byref(cfg), C.byref(bufs), st), fe),
                         ("dec", lambda: L.cs_lair_decode(hd, C.byref(cfg), C.byref(bufs), C.byref(rng), st), fd),
                         ("res", lambda: L.cs_lair_resample_impute(C.byref(cfg), C.byref(bufs), C.byref(rng), st), 0)):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"{name}: B={B} {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s  ({fl / ms / 1e9 / 157.3 * 100:.1f}% of f32 MFMA peak)")


def trace(B=65536):
    """Per-wave time line of one encoder and one decoder launch (clock64 stamps)."""
    import numpy as np
    D, Zd, H, K, R = 43, 32, 256, 4, 1
    enc_w = O.make_resfc_weights(D, 2 * Zd, 5, H, 1)
    dec_w = O.make_resfc_weights(Zd, 2 * D, 2, H, 2)
    enc = ResidualFCNetwork(D, 2 * Zd, 5, H).load_weight_dict(enc_w)
    dec = ResidualFCNetwork(Zd, 2 * D, 2, H).load_weight_dict(dec_w)
    model = VAE(dec, "normal_with_eps", enc, "normal_with_eps").cuda()
    X = torch.randn(B, K, D, device="cuda")
    M8 = (torch.rand(B, D, device="cuda") > 0.5).to(torch.uint8)
    ws = engine.lair_workspace(model, B, K, R, 1, X.device)
    cfg = engine.lair_cfg(model, ws, weighting="dmis", scheme="gr", method="multinomial", cluster_size=0, weight_alpha=None)
    bufs = _lib.make_bufs(X=X, M=M8, q=ws.q, Z=ws.Z, gen=ws.gen, gen_ll=ws.gen_ll)
    L = _lib.lib()
    L.cs_debug_set_trace.argtypes = [C.c_void_p]
    he, hd = model.var_latent_network.packed().handle, model.generator_network.packed().handle
    st = _lib.current_stream()
    rng = _lib.make_rng(1, 0, 0)
    for name, fn, rows, npts in (("enc", lambda: L.cs_lair_encode(he, C.byref(cfg), C.byref(bufs), st), B * K, 9),
                                 ("dec", lambda: L.cs_lair_decode(hd, C.byref(cfg), C.byref(bufs), C.byref(rng), st), B * (K + R), 6)):
        fn(); torch.cuda.synchronize()
        tiles = (rows + 31) // 32
        buf = torch.zeros(tiles, 16, dtype=torch.int64, device="cuda")
        L.cs_debug_set_trace(buf.data_ptr())
        fn(); torch.cuda.synchronize()
        L.cs_debug_set_trace(None)
        t = buf.cpu().numpy()[:, :npts].astype(np.float64)
        d = np.diff(t, axis=1)
        print(name, "tiles", tiles, "segment medians (ticks):", np.median(d, axis=0).astype(int).tolist(),
              "total median", int(np.median(t[:, -1] - t[:, 0])), "kernel span", int(t[:, -1].max() - t[:, 0].min()))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "trace":
        trace(int(sys.argv[2]) if len(sys.argv) > 2 else 65536)
        sys.exit(0)
    main()
