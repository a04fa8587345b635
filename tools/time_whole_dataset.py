"""End-to-end wall time of a whole-test-set configuration through the YAML driver, artefacts written as configured
(synthetic data of the dataset's size and shape).   python tools/time_whole_dataset.py <config.yaml> [overrides...]

With CS_PHASES=1 the time is also split by phase (each phase bracketed by a device synchronisation, which costs the
overlap between them: the total printed last is the one without the brackets only when CS_PHASES is unset)."""
import os
import sys
import time
from collections import defaultdict

import torch

sys.path.insert(0, ".")
from vae_conditional_sampling_amd import driver  # noqa: E402
from vae_conditional_sampling_amd.data import missing_data_module  # noqa: E402
from vae_conditional_sampling_amd.sampling import imputation_metrics  # noqa: E402
from vae_conditional_sampling_amd.utils import artifacts  # noqa: E402

phases = defaultdict(float)


def timed(name, fn):
    def wrapper(*a, **k):
        torch.cuda.synchronize()
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            torch.cuda.synchronize()
            phases[name] += time.perf_counter() - t
    return wrapper


if os.environ.get("CS_PHASES"):
    imputation_metrics.pointwise_summary = timed("epoch end: point-wise summaries", imputation_metrics.pointwise_summary)
    artifacts.AsyncNpzWriter.close = timed("epoch end: wait for the artefact writer", artifacts.AsyncNpzWriter.close)
    missing_data_module.MissingDataModule.setup = timed("set-up: data", missing_data_module.MissingDataModule.setup)
    driver.build_model = timed("set-up: model", driver.build_model)
    resolve = driver.resolve_test_class

    def resolve_timed(path):
        cls = resolve(path)
        if not getattr(cls, "_timed", False):
            cls.test_step = timed("batches: test_step", cls.test_step)
            cls._timed = True
        return cls
    driver.resolve_test_class = resolve_timed

cfg = driver.load_config(sys.argv[1], ["trainer.default_root_dir=/tmp/cs_whole_dataset"] + sys.argv[2:])
torch.cuda.synchronize()
t0 = time.perf_counter()
tester, outs = driver.run(cfg)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
a = cfg["test_class"]["init_args"]
T, K = a.get("num_iterations", a.get("num_mwg_iterations")), a["num_copies"]
N = tester.datamodule.num_test_total if hasattr(tester, "datamodule") else None
for name, sec in phases.items():
    print(f"  {name}: {sec:.2f} s")
if phases:
    print(f"  everything else: {dt - sum(phases.values()):.2f} s")
print(f"{sys.argv[1]}: {len(outs)} batches, N={N}, K={K}, T={T}: {dt:.2f} s end to end"
      + (f" = {N * K * T / dt:.3e} conditional samples/s including set-up, diagnostics and artefact files" if N else ""))
