import cProfile, pstats, sys, io
sys.path.insert(0, ".")
import torch
from vae_conditional_sampling_amd import driver
cfg = driver.load_config(sys.argv[1], ["trainer.default_root_dir=/tmp/cs_prof"] + sys.argv[2:])
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
driver.run(cfg)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:6000])
