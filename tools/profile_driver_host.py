"""cProfile of one driver run (host side): python tools/profile_driver_host.py <config.yaml> <iterations> [overrides ...]

Shows where the wall time of a YAML run goes outside the kernels: artefact compression, device-to-host copies of the
stored imputations, per-iteration diagnostics, ctypes calls."""
import cProfile, pstats, sys, time, io
sys.path.insert(0, ".")
import torch
from vae_conditional_sampling_amd import driver
cfg_path = sys.argv[1]; iters = int(sys.argv[2])
key = "num_iterations" if "lair" in cfg_path else "num_mwg_iterations"
cfg = driver.load_config(cfg_path, [f"test_class.init_args.{key}={iters}", "trainer.default_root_dir=/tmp/cs_prof_driver", "data.num_first_datapoints_test=1024"] + sys.argv[3:])
# warm
pr = cProfile.Profile()
torch.cuda.synchronize(); t0 = time.perf_counter()
pr.enable()
tester, outs = driver.run(cfg)
torch.cuda.synchronize()
pr.disable()
dt = time.perf_counter() - t0
print(f"{dt:.2f} s, {dt / (len(outs) * iters) * 1e3:.3f} ms per iteration")
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
