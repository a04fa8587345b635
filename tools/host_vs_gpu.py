"""Host time against GPU time of an observed LAIR run at a small batch: when the library call returns (all launches
issued) and when the GPU is done.   python tools/host_vs_gpu.py [B] [T]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from vae_conditional_sampling_amd import _lib  # noqa: E402
from vae_conditional_sampling_amd.models import VAE, ResidualFCNetwork  # noqa: E402
from vae_conditional_sampling_amd.sampling import engine, imputation_metrics as IM  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
T = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
w = bench.WORKLOADS["miniboone"]
D, Zd, H, K, R = w["D"], w["Zd"], w["H"], w["K"], w["R"]
dev = torch.device("cuda", 0)
torch.manual_seed(1)
model = VAE(ResidualFCNetwork(Zd, 2 * D, w["nbd"], H), "normal_with_eps",
            ResidualFCNetwork(D, 2 * Zd, w["nbe"], H), "normal_with_eps").to(dev)
X = torch.randn(B, K, D, device=dev).contiguous()
Xt = torch.randn(B, D, device=dev)
M = torch.rand(B, 1, D, device=dev) > 0.5
M8 = engine.mask_u8(M)
ws = engine.lair_workspace(model, B, K, R, 1, dev)
cfg = engine.lair_cfg(model, ws, weighting="dmis", scheme="gr", method="multinomial", cluster_size=0, weight_alpha=None)
q_true = IM.encode_posterior(model, Xt.unsqueeze(1))
for observed in (False, True):
    obs = None
    if observed:
        obs = engine.RunObserver(X_store=torch.empty(T, B, K, D, device=dev), X_true=Xt, rmse=torch.empty(T, B, K, device=dev),
                                 mae=torch.empty(T, B, K, device=dev), q_true=q_true, wass=torch.empty(T, B, K, device=dev))
    stats = torch.empty(3, T, B, device=dev)
    engine.lair_run(model, ws, X, M8, cfg, 50, rng=_lib.make_rng(1, 0, 0), stats_store=stats[:, :50], observe=obs.window(0, 50) if obs else None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    engine.lair_run(model, ws, X, M8, cfg, T, rng=_lib.make_rng(1, 0, 0), stats_store=stats, observe=obs)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"B={B} T={T} observed={observed}: call returned after {(t1 - t0) / T * 1e6:.1f} us per iteration, "
          f"GPU done after {(t2 - t0) / T * 1e6:.1f} us per iteration")
