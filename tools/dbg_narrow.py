import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cs_oracle as O
from vae_conditional_sampling_amd import _lib
from vae_conditional_sampling_amd.models import ResidualFCNetwork
_lib.set_mlp_arith("f16x2")
L = _lib.lib()
torch.set_printoptions(linewidth=200, precision=4)
in_dim, out_dim, nb, H, rows = 32, 32, 0, 64, 16
w = O.make_resfc_weights(in_dim, out_dim, nb, H, seed=1)
# first layer: hidden unit i = x_i (i < 32), else 0 ; output j = hidden j
w["W0"] = torch.zeros(H, in_dim); w["W0"][:in_dim, :in_dim] = torch.eye(in_dim)
w["b0"] = torch.zeros(H)
w["Wf"] = torch.zeros(out_dim, H); w["Wf"][:, :out_dim] = torch.eye(out_dim)
w["bf"] = torch.zeros(out_dim)
net = ResidualFCNetwork(in_dim, out_dim, nb, H).load_weight_dict(w).cuda().eval()
x = (torch.arange(rows * in_dim, dtype=torch.float32).reshape(rows, in_dim) / 100 + 1).cuda()
L.cs_debug_set_narrow(0); yw = net(x).clone()
L.cs_debug_set_narrow(1); yn = net(x).clone()
print("x row0", x[0]); print("wide row0", yw[0]); print("narrow row0", yn[0]); print("narrow row1", yn[1])
print("narrow col0", yn[:, 0])
d = (yn - yw)
print("diff row0", d[0, :8]); print("diff row15", d[15, :8]); print("x row15 max", float(x[15].max()))
# scale inputs by 8: does the error scale?
x2 = x * 8
L.cs_debug_set_narrow(0); yw2 = net(x2).clone()
L.cs_debug_set_narrow(1); yn2 = net(x2).clone()
print("x*8 diff row0", (yn2 - yw2)[0, :8])
# Wf = 2*I
w["Wf"] = w["Wf"] * 2
net2 = ResidualFCNetwork(in_dim, out_dim, nb, H).load_weight_dict(w).cuda().eval()
L.cs_debug_set_narrow(0); yw3 = net2(x).clone()
L.cs_debug_set_narrow(1); yn3 = net2(x).clone()
print("Wf*2 diff row0", (yn3 - yw3)[0, :8])
