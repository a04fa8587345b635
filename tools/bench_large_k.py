"""LAIR step with more than 64 proposals per example (MiniBooNE networks, K = 96 + 4): stage times through the C ABI.

    python tools/bench_large_k.py [examples] [steps]
"""
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    dev = torch.device("cuda:0")
    b = bench.LairCabi("miniboone_k100", B, 0, dev)
    el, evs = b.timed(steps, 3)
    roof, kern = b.kernel_report(evs, "f16x2")
    print(f"B={B} K=96+4: {el / steps * 1e3:.3f} ms per step, {B * 96 * steps / el:.3e} samples/s")
    print({k: round(v, 4) if isinstance(v, float) else v for k, v in kern.items()})


if __name__ == "__main__":
    main()
