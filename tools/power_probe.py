"""Sample GPU clock and power (rocm-smi) while the fused-MLP encoder kernel runs back to back."""
import ctypes as C, subprocess, sys, threading, time
import torch
sys.path.insert(0, ".")
from vae_conditional_sampling_amd import _lib
from vae_conditional_sampling_amd.models import VAE, ResidualFCNetwork
from vae_conditional_sampling_amd.sampling import engine

B, D, Zd, H, K, R = 65536, 43, 32, 256, 4, 1
enc = ResidualFCNetwork(D, 2 * Zd, 5, H)
dec = ResidualFCNetwork(Zd, 2 * D, 2, H)
model = VAE(dec, "normal_with_eps", enc, "normal_with_eps").cuda()
X = torch.randn(B, K, D, device="cuda")
M8 = (torch.rand(B, D, device="cuda") > 0.5).to(torch.uint8)
ws = engine.lair_workspace(model, B, K, R, 1, X.device)
cfg = engine.lair_cfg(model, ws, weighting="dmis", scheme="gr", method="multinomial", cluster_size=0, weight_alpha=None)
bufs = _lib.make_bufs(X=X, M=M8, q=ws.q, Z=ws.Z, gen=ws.gen, gen_ll=ws.gen_ll)
L = _lib.lib()
he = model.var_latent_network.packed().handle
st = _lib.current_stream()
stop = False
def sampler():
    while not stop:
        out = subprocess.run(["rocm-smi", "-c", "-P", "--showperflevel"], capture_output=True, text=True).stdout
        print("\n".join(l for l in out.splitlines() if "GPU[0]" in l or "clk" in l.lower() or "ower" in l), flush=True)
        time.sleep(0.5)
th = threading.Thread(target=sampler); th.start()
t0 = time.time()
n = 0
while time.time() - t0 < 6:
    for _ in range(50):
        L.cs_lair_encode(he, C.byref(cfg), C.byref(bufs), st)
    torch.cuda.synchronize(); n += 50
print("launches", n, "avg ms", (time.time() - t0) * 1e3 / n)
stop = True; th.join()
