"""CPU restatement of the reference's convolutional ResNet encoder / decoder.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the CPU-baseline leg of
bench.py; the product (vae_conditional_sampling_amd/) never imports it.

Plain torch-CPU float32 functional ops in the reference's order of operations, eval mode
(BatchNorm with running statistics, Dropout2d = identity):
  * encoder: ResNetEncoder.forward irwg/models/vae_resnet.py:128-149, EncoderBlock.forward :60-78,
    stage construction _make_layer :112-126;
  * decoder: ResNetDecoder.forward :259-277, DecoderBlock.forward :166-184, resize-convolutions
    :32-43, Interpolate :14-22, stage construction :243-257 and the constructor's resize rules
    :208-235.
Weights are a flat dict under the reference's state-dict key names plus a 'cfg' entry, so the same
dict loads into the reference modules (strict) in oracle/check_conv_against_reference.py, which
pins these functions bitwise against the reference, and into this repository's modules.
"""
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------
# random weights (torch's default initialisations, BatchNorm statistics randomised so that the
# normalisation is not the identity)
# ----------------------------------------------------------------------------------------
class _Init:
    def __init__(self, seed):
        self.g = torch.Generator().manual_seed(seed)
        self.sd = {}

    def u(self, *shape, bound):
        return ((torch.rand(*shape, generator=self.g) * 2 - 1) * bound).float()

    def conv(self, key, cout, cin, k):
        self.sd[key + ".weight"] = self.u(cout, cin, k, k, bound=1.0 / math.sqrt(cin * k * k))

    def linear(self, key, out, inp):
        b = 1.0 / math.sqrt(inp)
        self.sd[key + ".weight"] = self.u(out, inp, bound=b)
        self.sd[key + ".bias"] = self.u(out, bound=b)

    def bn(self, key, c):
        self.sd[key + ".weight"] = 0.5 + torch.rand(c, generator=self.g)
        self.sd[key + ".bias"] = self.u(c, bound=0.2)
        self.sd[key + ".running_mean"] = self.u(c, bound=0.2)
        self.sd[key + ".running_var"] = 0.5 + torch.rand(c, generator=self.g)
        self.sd[key + ".num_batches_tracked"] = torch.tensor(1, dtype=torch.int64)


def make_resnet_encoder_weights(input_shape, layers, layer_widths, latent_dim, seed):
    """latent_dim = network output width (2 Zd for a normal posterior)."""
    it = _Init(seed)
    cin = layer_widths[0]
    it.conv("conv1", cin, input_shape[0], 3)
    it.bn("bn1", cin)
    for i, (n, width) in enumerate(zip(layers, layer_widths)):
        for j in range(n):
            stride = 2 if (i > 0 and j == 0) else 1
            p = f"layers.{i}.{j}"
            it.conv(p + ".conv1", width, cin, 3); it.bn(p + ".bn1", width)
            it.conv(p + ".conv2", width, width, 3); it.bn(p + ".bn2", width)
            if j == 0 and (stride != 1 or cin != width):
                it.conv(p + ".downsample.0", width, cin, 1); it.bn(p + ".downsample.1", width)
            cin = width
    it.linear("final", latent_dim, cin)
    it.sd["cfg"] = dict(kind="resnet_enc", input_shape=tuple(input_shape), layers=list(layers),
                        layer_widths=list(layer_widths), latent_dim=latent_dim)
    return it.sd


def make_resnet_decoder_weights(output_shape, layers, layer_upscale, layer_widths, downscaled_resolution,
                                latent_dim, upscale_factor, seed):
    it = _Init(seed)
    cin = layer_widths[0] * 2
    it.linear("linear", cin * downscaled_resolution[0] * downscaled_resolution[1], latent_dim)
    for i, (n, width) in enumerate(zip(layers, layer_widths)):
        scale = 2 if i < len(layers) - 1 else 1
        for j in range(n):
            p = f"layers.{i}.{j}"
            sc = scale if j == 0 else 1
            it.conv(p + ".conv1", cin, cin, 3); it.bn(p + ".bn1", cin)
            it.conv(p + (".conv2.1" if sc != 1 else ".conv2"), width, cin, 3); it.bn(p + ".bn2", width)
            if j == 0 and (scale != 1 or cin != width):
                it.conv(p + (".upsample.0.1" if sc != 1 else ".upsample.0"), width, cin, 1)
                it.bn(p + ".upsample.1", width)
            cin = width
    it.conv("conv1", output_shape[0], cin, 3)
    it.sd["cfg"] = dict(kind="resnet_dec", output_shape=tuple(output_shape), layers=list(layers),
                        layer_upscale=list(layer_upscale), layer_widths=list(layer_widths),
                        downscaled_resolution=tuple(downscaled_resolution), latent_dim=latent_dim,
                        upscale_factor=upscale_factor)
    return it.sd


def state_dict_of(w):
    return {k: v for k, v in w.items() if k != "cfg"}


# ----------------------------------------------------------------------------------------
# forwards
# ----------------------------------------------------------------------------------------
def _bn(w, key, x):
    return F.batch_norm(x, w[key + ".running_mean"], w[key + ".running_var"], w[key + ".weight"], w[key + ".bias"],
                        False, 0.1, 1e-5)


def resnet_encoder_forward(w, x):
    cfg = w["cfg"]
    shape = x.shape
    nd = len(cfg["input_shape"])
    # leading dims: everything before the (flattened or shaped) datapoint dims (vae_resnet.py:143-147)
    numel = math.prod(cfg["input_shape"])
    lead, acc = None, 1
    for d in range(len(shape) - 1, -1, -1):
        acc *= shape[d]
        if acc == numel:  # the reference takes the LAST index whose trailing product matches
            lead = shape[:d]
            break
    assert lead is not None and nd >= 1
    h = x.reshape(-1, *cfg["input_shape"])
    h = F.relu(_bn(w, "bn1", F.conv2d(h, w["conv1.weight"], None, 1, 1)))
    for i, n in enumerate(cfg["layers"]):
        for j in range(n):
            p = f"layers.{i}.{j}"
            stride = 2 if (i > 0 and j == 0) else 1
            identity = h
            out = F.relu(_bn(w, p + ".bn1", F.conv2d(h, w[p + ".conv1.weight"], None, stride, 1)))
            out = _bn(w, p + ".bn2", F.conv2d(out, w[p + ".conv2.weight"], None, 1, 1))
            if p + ".downsample.0.weight" in w:
                identity = _bn(w, p + ".downsample.1", F.conv2d(h, w[p + ".downsample.0.weight"], None, stride, 0))
            h = F.relu(out + identity)
    h = torch.flatten(F.adaptive_avg_pool2d(h, (1, 1)), 1)
    h = h.reshape(*lead, -1)
    return F.linear(h, w["final.weight"], w["final.bias"])


def resnet_decoder_forward(w, z):
    cfg = w["cfg"]
    shape = z.shape
    h = F.linear(z.reshape(-1, shape[-1]), w["linear.weight"], w["linear.bias"])
    h = h.reshape(h.shape[0], cfg["layer_widths"][0] * 2, *cfg["downscaled_resolution"])
    h = F.interpolate(h, size=cfg["output_shape"][-2] // cfg["upscale_factor"])
    nl = len(cfg["layers"])
    for i, n in enumerate(cfg["layers"]):
        scale = 2 if i < nl - 1 else 1
        for j in range(n):
            p = f"layers.{i}.{j}"
            sc = scale if j == 0 else 1
            identity = h
            out = F.relu(_bn(w, p + ".bn1", F.conv2d(h, w[p + ".conv1.weight"], None, 1, 1)))
            if sc != 1:
                out = F.conv2d(F.interpolate(out, scale_factor=sc), w[p + ".conv2.1.weight"], None, 1, 1)
            else:
                out = F.conv2d(out, w[p + ".conv2.weight"], None, 1, 1)
            out = _bn(w, p + ".bn2", out)
            if p + ".upsample.0.1.weight" in w:
                identity = _bn(w, p + ".upsample.1",
                               F.conv2d(F.interpolate(h, scale_factor=sc), w[p + ".upsample.0.1.weight"], None, 1, 0))
            elif p + ".upsample.0.weight" in w:
                identity = _bn(w, p + ".upsample.1", F.conv2d(h, w[p + ".upsample.0.weight"], None, 1, 0))
            h = F.relu(out + identity)
        up = cfg["layer_upscale"][i]
        if up is not None:
            h = F.interpolate(h, size=up)
    h = F.interpolate(h, scale_factor=1)  # the reference's trailing Interpolate(scale_factor=1), an identity
    h = F.conv2d(h, w["conv1.weight"], None, 1, 1)
    return h.reshape(*shape[:-1], -1)


def net_forward(w, x):
    """Dispatch on the weight dict: residual MLP (cs_oracle.resfc_forward) or ResNet."""
    kind = w.get("cfg", {}).get("kind") if isinstance(w.get("cfg"), dict) else None
    if kind == "resnet_enc":
        return resnet_encoder_forward(w, x)
    if kind == "resnet_dec":
        return resnet_decoder_forward(w, z=x)
    from . import cs_oracle
    return cs_oracle.resfc_forward(w, x)
