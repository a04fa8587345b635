"""Time the REFERENCE package's own CPU path for the benchmarked workloads (build container only).

    python tools/time_reference_cpu.py [--threads N] [--out profiles/<name>.json]

Imports the reference from /root/reference through oracle/ref_harness.py (test infrastructure; the reference does
not exist on the GPU box, so this is not part of bench.py), builds random-init VAEs of the configurations'
architectures and times, with torch's CPU float32 ops and its own generator:
  * one LAIR iteration, importance_resampling_gibbs_iteration (irwg/sampling/vae_imputation.py:484-726), at the
    MiniBooNE (B=512, K=4, R=1), HEPMASS (B=256, K=16, R=1) and GAS (B=4096, K=16, R=1) shapes;
  * one AC-MWG step through TestVAEAdaptiveCollapsedMetropolisWithinGibbs.sampler
    (irwg/sampling/test_step_vae_sampling.py:1664-1768) at the MiniBooNE shape (B=256, K=5, p=0.05).
Median of the steady-state iterations after two warm-up iterations; conditional samples/s = B*K / seconds.
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.ref_harness import load_reference  # noqa: E402

SHAPES = {
    "miniboone": dict(D=43, Zd=32, H=256, nbe=5, nbd=2, B=512, K=4, R=1),
    "hepmass": dict(D=21, Zd=16, H=256, nbe=2, nbd=2, B=256, K=16, R=1),
    "gas": dict(D=8, Zd=16, H=256, nbe=2, nbd=2, B=4096, K=16, R=1),
}
BASE_KW = dict(store_last_n_iterations=-1, store_every_n_iterations=1, imputation_metric_quantiles=[],
               compute_f1_metric=False, compute_inv_f1_metric=False, compute_bin_accuracy_metric=False,
               compute_rmse_metric=False, compute_ssim_metric=False, compute_latent_wass_metric=False,
               imputation_fn="vae_samples")


def ref_vae(ns, s):
    torch.manual_seed(20220118)
    enc = ns.neural_nets.ResidualFCNetwork(s["D"], 2 * s["Zd"], s["nbe"], s["H"])
    dec = ns.neural_nets.ResidualFCNetwork(s["Zd"], 2 * s["D"], s["nbd"], s["H"])
    Dn = ns.vae.DISTRIBUTION
    return ns.vae.VAE(dec, Dn["normal_with_eps"], enc, Dn["normal_with_eps"], num_latent_samples=1, lr_generator=1e-3,
                      lr_latent=1e-3).eval()


def inputs(B, K, D):
    g = torch.Generator().manual_seed(7)
    X_true = torch.randn(B, D, generator=g)
    M = torch.rand(B, D, generator=g) > 0.5
    M[:, 0] = True
    X = X_true.unsqueeze(1).repeat(1, K, 1)
    X = X * M.unsqueeze(1) + torch.randn(B, K, D, generator=g) * (~M.unsqueeze(1))
    return X.contiguous(), M.unsqueeze(1).contiguous()


def time_lair(ns, name, iters):
    s = SHAPES[name]
    vae = ref_vae(ns, s)
    X, M = inputs(s["B"], s["K"], s["D"])
    times = []
    with torch.no_grad():
        for t in range(iters + 2):
            t0 = time.perf_counter()
            X = ns.vae_imputation.importance_resampling_gibbs_iteration(
                vae, X, M, 1, glr_cluster_size=None, weighting="dmis", resampling_method="multinomial",
                resampling_scheme="gr", num_prior_replenish_proposals=s["R"])[0]
            times.append(time.perf_counter() - t0)
    sec = statistics.median(times[2:])
    return {"what": f"LAIR iteration, {name} shape, B={s['B']} K={s['K']} R={s['R']}", "ms_per_iteration": sec * 1e3,
            "samples_per_s": s["B"] * s["K"] / sec, "iterations": iters}


def time_acmwg(ns, iters, B=256, K=5):
    s = SHAPES["miniboone"]
    vae = ref_vae(ns, s)
    X, M = inputs(B, K, s["D"])
    tester = ns.test_step.TestVAEAdaptiveCollapsedMetropolisWithinGibbs(
        num_pseudo_warmup_iterations=0, num_mwg_iterations=iters + 2, prior_mixture_probability=0.05, warmup_method="none",
        num_copies=K, **BASE_KW)
    tester.set_model(vae)
    times = []
    with torch.no_grad():
        t0 = time.perf_counter()
        for _ in tester.sampler(X, M):
            t1 = time.perf_counter()
            times.append(t1 - t0)
            t0 = t1
    sec = statistics.median(times[2:])
    return {"what": f"AC-MWG step, miniboone shape, B={B} K={K} p=0.05", "ms_per_iteration": sec * 1e3,
            "samples_per_s": B * K / sec, "iterations": iters}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    ns = load_reference()
    res = {"host": f"build container, {os.cpu_count()} CPUs, torch {torch.__version__} CPU float32, {args.threads} threads",
           "source": "the reference package itself (/root/reference), imported through oracle/ref_harness.py",
           "lair_miniboone": time_lair(ns, "miniboone", args.iters), "lair_hepmass": time_lair(ns, "hepmass", args.iters),
           "lair_gas": time_lair(ns, "gas", max(5, args.iters // 4)), "acmwg_miniboone": time_acmwg(ns, args.iters)}
    text = json.dumps(res, indent=1)
    print(text)
    if args.out:
        with open(args.out, "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
