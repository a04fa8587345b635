import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from vae_conditional_sampling_amd import _lib
from vae_conditional_sampling_amd.models import VAE, ResidualFCNetwork
from vae_conditional_sampling_amd.sampling import engine
B = 65536
D, Zd, H, K, R = 43, 32, 256, 4, 1
enc = ResidualFCNetwork(D, 2 * Zd, 5, H); dec = ResidualFCNetwork(Zd, 2 * D, 2, H)
model = VAE(dec, "normal_with_eps", enc, "normal_with_eps").cuda()
X = torch.randn(B, K, D, device="cuda"); M8 = (torch.rand(B, D, device="cuda") > 0.5).to(torch.uint8)
ws = engine.lair_workspace(model, B, K, R, 1, X.device)
cfg = engine.lair_cfg(model, ws, weighting="dmis", scheme="gr", method="multinomial", cluster_size=0, weight_alpha=None)
bufs = _lib.make_bufs(X=X, M=M8, q=ws.q, Z=ws.Z, gen=ws.gen, gen_ll=ws.gen_ll)
L = _lib.lib(); L.cs_debug_set_trace.argtypes = [C.c_void_p]
he = model.var_latent_network.packed().handle
st = _lib.current_stream()
fn = lambda: L.cs_lair_encode(he, C.byref(cfg), C.byref(bufs), st)
fn(); fn(); torch.cuda.synchronize()
tiles = B * K // 16
buf = torch.zeros(tiles, 16, dtype=torch.int64, device="cuda")
L.cs_debug_set_trace(buf.data_ptr()); fn(); torch.cuda.synchronize(); L.cs_debug_set_trace(None)
np.save("gpurun_out/trace_enc.npy", buf.cpu().numpy())
