"""What the vendor GEMM library sustains on this box (torch.matmul -> hipBLASLt / rocBLAS), as a yardstick for the
hand-written kernels: dense bf16 and f32 products of random normal matrices, shader clock settled first.
Prints one JSON line.  python tools/micro/library_gemm.py [n]"""
import json
import sys
import time

import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda")
out = {"n": n}
for name, dt, reps in (("bf16", torch.bfloat16, 60), ("f32", torch.float32, 12)):
    a = torch.randn(n, n, device=dev).to(dt)
    b = torch.randn(n, n, device=dev).to(dt)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:      # settle the clock
        a @ b
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        a @ b
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    out[f"{name}_ms"] = ms
    out[f"{name}_tflops"] = 2 * n ** 3 / ms / 1e9
print(json.dumps(out))
