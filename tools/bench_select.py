import sys, time, torch
sys.path.insert(0, ".")
from vae_conditional_sampling_amd.sampling.imputation_metrics import _order_statistics
x = torch.rand(47, 2799792, device="cuda")
m = x.shape[1]
ranks = sorted({int(q * (m - 1)) for q in (0.1, 0.25, 0.5, 0.75, 0.9)} | {int(q * (m - 1)) + 1 for q in (0.1, 0.25, 0.5, 0.75, 0.9)})
for _ in range(2):
    _order_statistics(x, ranks)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5):
    _order_statistics(x, ranks)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
print(f"select: {dt*1e3:.2f} ms for {x.numel()*4/1e9:.2f} GB = {3*x.numel()*4/dt/1e12:.2f} TB/s over three passes")
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(3):
    s = x.sort(dim=1).values
torch.cuda.synchronize(); print(f"sort: {(time.perf_counter()-t)/3*1e3:.2f} ms")
