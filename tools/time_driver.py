"""Wall time per sampler iteration of a YAML configuration through the driver (host overhead included)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from vae_conditional_sampling_amd import driver  # noqa: E402

cfg_path, iters = sys.argv[1], int(sys.argv[2])
over = sys.argv[3:]
key = "num_iterations" if "lair" in cfg_path else "num_mwg_iterations"
cfg = driver.load_config(cfg_path, [f"test_class.init_args.{key}={iters}", "trainer.default_root_dir=/tmp/cs_time_driver",
                                    "test_class.init_args.dont_store_imputations=True"] + over)
torch.cuda.synchronize()
t0 = time.perf_counter()
tester, outs = driver.run(cfg)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
nb = len(outs)
print(f"{cfg_path}: {nb} batches, {dt:.2f} s total, {dt / (nb * iters) * 1e3:.3f} ms per iteration (incl. set-up, diagnostics, storing)")
