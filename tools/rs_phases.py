#!/usr/bin/env python3
"""Debug tool: mean cycles per phase of cs_lair_resample_kernel, per wavefront.

Needs a library whose cs_lair.hip was compiled with -DCS_RS_PROFILE (the kernel then overwrites
ess[] with the phase durations of each wavefront); not part of the product build.

    CSVAE_EXP_LIB=<that library> python tools/rs_phases.py [miniboone|hepmass|gas] [B] [examples per wavefront]
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "miniboone"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    from vae_conditional_sampling_amd import _lib
    if os.environ.get("CSVAE_EXP_LIB"):   # the library built with -DCS_RS_PROFILE (this tool only)
        _lib.LIB_PATH = os.path.abspath(os.environ["CSVAE_EXP_LIB"])
    from vae_conditional_sampling_amd.models import VAE, ResidualFCNetwork
    from vae_conditional_sampling_amd.sampling import engine
    w = bench.WORKLOADS[name]
    D, Zd, H, K, R = w["D"], w["Zd"], w["H"], w["K"], w["R"]
    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    model = VAE(ResidualFCNetwork(Zd, 2 * D, w["nbd"], H), "normal_with_eps",
                ResidualFCNetwork(D, 2 * Zd, w["nbe"], H), "normal_with_eps").to(dev)
    M = torch.rand(B, D) > 0.5
    X = torch.randn(B, K, D).to(dev).contiguous()
    M8 = M.to(torch.uint8).to(dev).contiguous()
    ws = engine.lair_workspace(model, B, K, R, 1, dev)
    cfg = engine.lair_cfg(model, ws, weighting="dmis", scheme="gr", method="multinomial", cluster_size=0, weight_alpha=None)
    bufs = _lib.make_bufs(X=X, M=M8, q=ws.q, Z=ws.Z, gen=ws.gen, gen_ll=ws.gen_ll, log_w=ws.log_w, norm_log_w=ws.norm_log_w,
                          idx=ws.idx, ess=ws.stats[0], tess=ws.stats[1], perplexity=ws.stats[2])
    L = _lib.lib()
    stream = _lib.current_stream()
    for t in range(3):
        rng = _lib.make_rng(5, t, 0)
        _lib.check(L.cs_lair_encode(model.var_latent_network.packed().handle, C.byref(cfg), C.byref(bufs), stream))
        _lib.check(L.cs_lair_decode(model.generator_network.packed().handle, C.byref(cfg), C.byref(bufs), C.byref(rng), stream))
        _lib.check(L.cs_lair_resample_impute(C.byref(cfg), C.byref(bufs), C.byref(rng), stream))
    torch.cuda.synchronize()
    E = int(sys.argv[3]) if len(sys.argv) > 3 else 64 // (K + R)   # examples per wavefront chosen by the library
    n = (B // E) * E
    per = 3 if E >= 3 else 2          # phase durations per statistics array (see the kernel's profile block)
    ph = []
    for i in range(7 if E >= 3 else 6):
        ph.append(float(ws.stats[i // per][:n].view(-1, E)[:, i % per].double().mean()))
    names = ["stage q/Z", "density + log_w", "ess/perplexity", "group cdf", "draws", "idx/latents out", "impute"]
    tot = sum(ph)
    for nm, v in zip(names, ph):
        print(f"{nm:18s} {v:10.0f} cycles  {100 * v / tot:5.1f} %")
    print(f"{'total':18s} {tot:10.0f}  (E = {E} examples per wavefront)")


if __name__ == "__main__":
    main()
