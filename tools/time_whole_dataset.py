"""End-to-end wall time of a whole-test-set configuration through the YAML driver, artefacts written as configured
(synthetic data of the dataset's size and shape).   python tools/time_whole_dataset.py <config.yaml> [overrides...]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from vae_conditional_sampling_amd import driver  # noqa: E402

cfg = driver.load_config(sys.argv[1], ["trainer.default_root_dir=/tmp/cs_whole_dataset"] + sys.argv[2:])
torch.cuda.synchronize()
t0 = time.perf_counter()
tester, outs = driver.run(cfg)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
a = cfg["test_class"]["init_args"]
T, K = a.get("num_iterations", a.get("num_mwg_iterations")), a["num_copies"]
N = tester.datamodule.num_test_total if hasattr(tester, "datamodule") else None
print(f"{sys.argv[1]}: {len(outs)} batches, N={N}, K={K}, T={T}: {dt:.2f} s end to end"
      + (f" = {N * K * T / dt:.3e} conditional samples/s including set-up, diagnostics and artefact files" if N else ""))
