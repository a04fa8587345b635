#!/usr/bin/env python3
"""Micro-benchmark of the fused MLP kernel (encoder / decoder launches of the LAIR step)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vae_conditional_sampling_amd import _lib  # noqa: E402

if os.environ.get("CSVAE_LIB_PATH"):   # timing experiments on a differently built library (this tool only)
    _lib.LIB_PATH = os.path.abspath(os.environ["CSVAE_LIB_PATH"])
from vae_conditional_sampling_amd.models import VAE, ResidualFCNetwork  # noqa: E402
from vae_conditional_sampling_amd.sampling import engine  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    D, Zd, H, K, R = 43, 32, 256, 4, 1
    enc = ResidualFCNetwork(D, 2 * Zd, 5, H)
    dec = ResidualFCNetwork(Zd, 2 * D, 2, H)
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
    enc = ResidualFCNetwork(D, 2 * Zd, 5, H)
    dec = ResidualFCNetwork(Zd, 2 * D, 2, H)
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
    for name, fn, rows, npts in (("enc", lambda: L.cs_lair_encode(he, C.byref(cfg), C.byref(bufs), st), B * K, int(os.environ.get("CS_TRACE_ENC_PTS", 10))),
                                 ("dec", lambda: L.cs_lair_decode(hd, C.byref(cfg), C.byref(bufs), C.byref(rng), st), B * (K + R), int(os.environ.get("CS_TRACE_DEC_PTS", 7)))):
        fn(); torch.cuda.synchronize()
        tiles = (rows + 15) // 16
        buf = torch.zeros(tiles, 16, dtype=torch.int64, device="cuda")
        L.cs_debug_set_trace(buf.data_ptr())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        L.cs_debug_set_trace(None)
        t = buf.cpu().numpy()[:, :npts].astype(np.float64)
        t = t[t[:, 0] != 0]          # the split-bf16 kernel stamps one record per 32 rows
        d = np.diff(t, axis=1)
        print(name, "tiles", tiles, "segment medians (ticks):", np.median(d, axis=0).astype(int).tolist(),
              "total median", int(np.median(t[:, -1] - t[:, 0])))
        # occupancy time line per XCD (the tick counters of different XCDs are not aligned)
        raw = buf.cpu().numpy()
        raw = raw[raw[:, 0] != 0]
        hw = raw[:, 15]
        xcc = (hw >> 32) & 0xF
        hwid = hw & 0xFFFFFFFF
        cu = (hwid >> 8) & 0xF; sh = (hwid >> 12) & 0x1; se = (hwid >> 13) & 0x7; simd = (hwid >> 4) & 0x3
        t0, t1 = raw[:, 0].astype(np.float64), raw[:, npts - 1].astype(np.float64)
        print(name, f"launch {ms:.3f} ms = {ms * 2.39e6:.0f} ticks at 2.39 GHz")
        # per CU (tick counters of different XCDs / shader engines are not aligned)
        key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
        spans, busy = [], []
        shown = 0
        for k_ in sorted(set(key.tolist())):
            sel = key == k_
            base = t0[sel].min()
            a0, a1 = t0[sel] - base, t1[sel] - base
            spans.append(a1.max())
            busy.append((a1 - a0).sum() / 8.0 / a1.max())
            if shown < 2:
                shown += 1
                edges = np.linspace(0, a1.max(), 21)
                occ = [int(((a0 <= e) & (a1 > e)).sum()) for e in edges[:-1]]
                print(f"  cu key {k_}: {sel.sum()} waves, span {a1.max():.0f} ticks, resident waves at 20 points: {occ}")
        spans, busy = np.array(spans), np.array(busy)
        print(f"  {len(spans)} CUs: span median {np.median(spans):.0f} min {spans.min():.0f} max {spans.max():.0f} ticks;",
              f"mean residency {busy.mean() * 8:.2f} of 8 wave slots; launch/span tick rate {np.median(spans) / (ms * 1e6):.3f} GHz")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "trace":
        trace(int(sys.argv[2]) if len(sys.argv) > 2 else 65536)
        sys.exit(0)
    main()
