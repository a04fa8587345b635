"""AC-MWG throughput through the sampler class (counter RNG, no warm-up).

    python tools/bench_acmwg.py [miniboone|omniglot] [examples] [iterations]

Reports chain-iterations (= conditional samples) per second over the MWG iterations, steady state.
"""
import sys
import time

import torch

sys.path.insert(0, ".")
from vae_conditional_sampling_amd.models import VAE, ResidualFCNetwork, ResNetDecoder, ResNetEncoder  # noqa: E402
from vae_conditional_sampling_amd.sampling.test_step_vae_sampling import \
    TestVAEAdaptiveCollapsedMetropolisWithinGibbs as Sampler  # noqa: E402

BASE_KW = dict(store_last_n_iterations=-1, store_every_n_iterations=1, imputation_metric_quantiles=[],
               compute_f1_metric=False, compute_inv_f1_metric=False, compute_bin_accuracy_metric=False,
               compute_rmse_metric=False, compute_ssim_metric=False, compute_latent_wass_metric=False,
               imputation_fn="vae_samples")


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "miniboone"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    K = 5
    torch.manual_seed(0)
    if name == "miniboone":
        D = 43
        model = VAE(ResidualFCNetwork(32, 86, 2, 256), "normal_with_eps", ResidualFCNetwork(43, 64, 5, 256), "normal_with_eps")
        X = torch.randn(B, D)
        M = torch.rand(B, D) > 0.5
        flops = 1365504 + 584704
    else:
        D = 784
        enc = ResNetEncoder((1, 28, 28), [2, 2, 2, 2], [64, 128, 256, 512], 100)
        dec = ResNetDecoder((1, 28, 28), [2, 2, 2, 2], [7, None, None, None], [256, 128, 64, 64], (4, 4), 50, 7)
        model = VAE(dec, "bern", enc, "normal")
        X = (torch.rand(B, D) < 0.2).float()
        M = torch.ones(B, 28, 28, dtype=torch.bool)
        M[:, :14, :14] = False
        M[:, 14:, 14:] = False
        M = M.reshape(B, D)
        flops = 911693824 + 1230915584
    M[:, 0] = True
    model = model.cuda()
    s = Sampler(num_pseudo_warmup_iterations=0, num_mwg_iterations=T, prior_mixture_probability=0.05, warmup_method="none",
                num_copies=K, **BASE_KW)
    s.set_model(model)
    s.on_test_start()
    Xk = (X * M).unsqueeze(1).repeat(1, K, 1).cuda()
    Mk = M.unsqueeze(1).cuda()
    X0 = s.basic_imputation(Xk, Mk)
    skip = max(2, T // 10)
    acc = []
    for t, o in enumerate(s.sampler(X0, Mk)):
        if t == skip:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        acc.append(o["acceptance"])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = T - skip
    from vae_conditional_sampling_amd import _lib
    ARITH = _lib.get_mlp_arith()
    PEAK = 157.3 if ARITH == "f32" else 2500.0 / 6    # algorithmic TFLOP/s: exact f32, or dense bf16 peak / 6 term products
    print(f"{name} AC-MWG: B={B} K={K} {n} iterations {dt / n * 1e3:.3f} ms/iter  {B * K * n / dt:.3e} samples/s  "
          f"{B * K * n * flops / dt / 1e12:.1f} algorithmic TFLOP/s ({B * K * n * flops / dt / 1e12 / PEAK * 100:.1f}% of the {ARITH} ceiling {PEAK:.1f})  "
          f"mean acceptance {float(torch.stack([a if torch.is_tensor(a) else torch.tensor(a) for a in acc[skip:]]).float().mean()):.3f}")


if __name__ == "__main__":
    main()
