"""One YAML configuration through the driver under each float32-equivalent arithmetic of the fused MLP kernel, same
seeds (the counter RNG gives every run the same draws): how far do whole runs drift apart?

    python tools/compare_arith_end_to_end.py <config.yaml> [iterations] [overrides ...]

A chain follows the exact-float32 run bit for bit in its indices until a resampling (or accept) decision falls
within rounding distance of its uniform; from then on it is a different, equally valid, realisation.  Printed per
arithmetic against the exact-float32 run: the share of chains whose last-iteration imputation still agrees to 1e-4,
and the largest deviation of every per-iteration curve of metrics_summary.npz relative to the curve's mean magnitude.
"""
import glob
import os
import shutil
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from vae_conditional_sampling_amd import driver  # noqa: E402


def run(cfg_path, iters, arith, over):
    root = f"/tmp/cs_cmp_arith_{arith}"
    shutil.rmtree(root, ignore_errors=True)
    key = "num_iterations" if "lair" in cfg_path else "num_mwg_iterations"
    cfg = driver.load_config(cfg_path, [f"test_class.init_args.{key}={iters}", f"trainer.default_root_dir={root}",
                                        "test_class.init_args.store_last_n_iterations=1", f"mlp_arith={arith}"] + over)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tester, _ = driver.run(cfg)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    d = tester.out_dir
    imps = [np.load(f)["imputations"] for f in sorted(glob.glob(os.path.join(d, "imputations_*.npz")),
                                                      key=lambda p: int(p.rsplit("_", 1)[1][:-4]))]
    last = np.concatenate([a[-1] for a in imps], axis=0)          # [N, K, D] of the last stored iteration
    summ = dict(np.load(os.path.join(d, "metrics_summary.npz")))
    return last, summ, dt


def main():
    cfg_path = sys.argv[1]
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    over = sys.argv[3:]
    ref_last, ref_summ, dt = run(cfg_path, iters, "f32", over)
    print(f"{cfg_path}, {iters} iterations, {ref_last.shape[0]} examples x {ref_last.shape[1]} chains; exact f32 run {dt:.1f} s")
    for arith in ("bf16x3", "f16x2"):
        last, summ, dt = run(cfg_path, iters, arith, over)
        scale = np.abs(ref_last).max(axis=(0, 1), keepdims=True) + 1e-30
        same = (np.abs(last - ref_last) / scale).max(axis=2) < 1e-4
        print(f"{arith}: {dt:.1f} s; chains whose last imputation agrees with the f32 run to 1e-4: "
              f"{same.mean() * 100:.2f} % ({int((~same).sum())} of {same.size} differ); mean |difference| over all entries / data scale: "
              f"{float((np.abs(last - ref_last) / scale).mean()):.3e}")
        worst = []
        for k, v in sorted(ref_summ.items()):
            w = summ.get(k)
            if w is None or w.shape != v.shape or v.dtype.kind != "f" or v.size == 0:
                continue
            ok = np.isfinite(v) & np.isfinite(w)
            if not ok.any():
                continue
            mag = float(np.abs(v[ok]).mean()) or 1.0
            worst.append((float(np.abs(v[ok] - w[ok]).max() / mag), k))
        worst.sort(reverse=True)
        print("   largest deviations of the per-iteration curves (share of the curve's mean magnitude): " +
              ", ".join(f"{k} {d:.2e}" for d, k in worst[:6]))


if __name__ == "__main__":
    main()
