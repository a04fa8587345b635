"""Per-phase tick breakdown of the fused MLP kernel launches of a LAIR step (encoder and decoder), from the
clock stamps the kernel writes when a trace buffer is set (cs_debug_set_trace): per 16-row tile
[start, prologue, first layer, block 0.., output layer, epilogue].  Prints the median duration of every phase and
the share of a workgroup's wall time each one takes.  Usage: python tools/trace_phases.py [B]"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import os  # noqa: E402

from vae_conditional_sampling_amd import _lib  # noqa: E402

if os.environ.get("CSVAE_EXP_LIB"):  # an experimental build of the library (timing experiments)
    _lib.LIB_PATH = os.path.abspath(os.environ["CSVAE_EXP_LIB"])
from vae_conditional_sampling_amd.models import VAE, ResidualFCNetwork  # noqa: E402
from vae_conditional_sampling_amd.sampling import engine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
D, Zd, H, K, R = 43, 32, 256, 4, 1
enc = ResidualFCNetwork(D, 2 * Zd, 5, H)
dec = ResidualFCNetwork(Zd, 2 * D, 2, H)
model = VAE(dec, "normal_with_eps", enc, "normal_with_eps").cuda()
X = torch.randn(B, K, D, device="cuda")
M8 = (torch.rand(B, D, device="cuda") > 0.5).to(torch.uint8)
ws = engine.lair_workspace(model, B, K, R, 1, X.device)
cfg = engine.lair_cfg(model, ws, weighting="dmis", scheme="gr", method="multinomial", cluster_size=0, weight_alpha=None)
bufs = _lib.make_bufs(X=X, M=M8, q=ws.q, Z=ws.Z, gen=ws.gen, gen_ll=ws.gen_ll)
rng = _lib.make_rng(0)
L = _lib.lib()
L.cs_debug_set_trace.argtypes = [C.c_void_p]
he = model.var_latent_network.packed().handle
hd = model.generator_network.packed().handle
st = _lib.current_stream()
legs = {
    "encoder": (lambda: L.cs_lair_encode(he, C.byref(cfg), C.byref(bufs), st), B * K, 5),
    "decoder": (lambda: L.cs_lair_decode(hd, C.byref(cfg), C.byref(bufs), C.byref(rng), st), B * (K + R), 2),
}
# let the shader clock settle
t = torch.randn(4096, 4096, device="cuda")
for _ in range(60):
    t @ t
torch.cuda.synchronize()
for name, (fn, rows, nb) in legs.items():
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"== {name}: {e0.elapsed_time(e1) / 20:.4f} ms per launch")
    tiles = (rows + 15) // 16
    buf = torch.zeros(tiles, 16, dtype=torch.int64, device="cuda")
    L.cs_debug_set_trace(buf.data_ptr())
    fn()
    torch.cuda.synchronize()
    L.cs_debug_set_trace(None)
    tr = buf.cpu().numpy()
    nst = 4 + nb + 1  # start, prologue, first layer, nb blocks, output layer, epilogue
    names = ["prologue", "first layer"] + [f"block {i}" for i in range(nb)] + ["output layer", "epilogue"]
    d = np.diff(tr[:, :nst], axis=1).astype(np.float64)
    tot = (tr[:, nst - 1] - tr[:, 0]).astype(np.float64)
    print(f"== {name}: {tiles} tiles, kernel span {(tr[:, :nst].max() - tr[:, 0].min())} ticks, "
          f"median tile life {np.median(tot):.0f} ticks")
    for i, n in enumerate(names):
        print(f"  {n:14s} median {np.median(d[:, i]):9.0f}  mean {d[:, i].mean():9.0f}  share {d[:, i].sum() / tot.sum():6.1%}")
    np.save(f"gpurun_out/trace_{name}.npy", tr)
