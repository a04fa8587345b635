"""Convolutional ResNet encoder / decoder with the reference's constructors and state-dict
layout, executed as layer programs by the HIP library.

Mirrors ``ResNetEncoder`` (irwg/models/vae_resnet.py:80-149, blocks :45-78) and
``ResNetDecoder`` (:186-277, blocks :151-184, resize-convolutions :32-43): parameters and
BatchNorm buffers are registered under the same names (``conv1``, ``bn1``,
``layers.{i}.{j}.conv1|bn1|conv2|bn2``, ``downsample.0|1``, ``final``; decoder ``linear``,
``conv2.1`` / ``upsample.0.1`` for the resize-convolutions, ``conv1``), so the
``var_latent_network.*`` / ``generator_network.*`` entries of a Lightning checkpoint load
unchanged.  The modules hold parameters only: ``forward`` runs the packed program
(eval mode: BatchNorm folded into the convolutions, Dropout2d = identity).
"""
from math import prod
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from .. import _lib


class _Resize(nn.Module):
    """Parameter-free placeholder of the reference's Interpolate: keeps the index it occupies in
    the surrounding nn.Sequential, so that the convolution after it is entry ``.1``."""

    def __init__(self, scale):
        super().__init__()
        self.scale = scale


def _conv(cin, cout, k, stride=1):
    return nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=(k - 1) // 2, bias=False)


def _resize_conv(cin, cout, k, scale):
    return _conv(cin, cout, k) if scale == 1 else nn.Sequential(_Resize(scale), _conv(cin, cout, k))


def _conv_of(m):
    """(convolution module, nearest-upsampling factor in front of it)."""
    return (m[1], m[0].scale) if isinstance(m, nn.Sequential) else (m, 1)


def _fold(conv, bn):
    """Eval-mode BatchNorm folded into the bias-free convolution before it."""
    w = conv.weight.detach().double().cpu()
    s = bn.weight.detach().double().cpu() / torch.sqrt(bn.running_var.detach().double().cpu() + bn.eps)
    b = bn.bias.detach().double().cpu() - bn.running_mean.detach().double().cpu() * s
    return (w * s.view(-1, 1, 1, 1)).float(), b.float()


class _EncBlock(nn.Module):
    def __init__(self, cin, cout, stride, shortcut):
        super().__init__()
        self.conv1, self.bn1 = _conv(cin, cout, 3, stride), nn.BatchNorm2d(cout)
        self.conv2, self.bn2 = _conv(cout, cout, 3), nn.BatchNorm2d(cout)
        self.downsample = nn.Sequential(_conv(cin, cout, 1, stride), nn.BatchNorm2d(cout)) if shortcut else None
        self.stride = stride


class _DecBlock(nn.Module):
    def __init__(self, cin, cout, scale, shortcut):
        super().__init__()
        self.conv1, self.bn1 = _conv(cin, cin, 3), nn.BatchNorm2d(cin)
        self.conv2, self.bn2 = _resize_conv(cin, cout, 3, scale), nn.BatchNorm2d(cout)
        self.upsample = nn.Sequential(_resize_conv(cin, cout, 1, scale), nn.BatchNorm2d(cout)) if shortcut else None


class _ProgramNet(nn.Module):
    out_split = 2

    def _init_cache(self):
        self._packed = None
        self._packed_key = None
        self._tensors = None
        self.requires_grad_(False)

    def packed(self):
        """Device handle of the layer program (built lazily, rebuilt when weights change)."""
        if self._tensors is None:  # gathered once: module traversal is the expensive part of this check
            self._tensors = list(self.parameters()) + list(self.buffers())
        key = (tuple((t.data_ptr(), t._version) for t in self._tensors), self.out_split, torch.cuda.current_device())
        if self._packed is None or self._packed_key != key:
            self._packed = self._program().build()
            self._packed_key = key
        return self._packed

    def forward(self, x):
        if self.training:
            raise NotImplementedError("training mode (batch statistics, dropout) is not implemented; call .eval()")
        return self.packed().forward(x)


def _scratch(*used):
    """Scratch buffer ids (1..4) not in `used`."""
    return [i for i in (1, 2, 3, 4) if i not in used]


class ResNetEncoder(_ProgramNet):
    def __init__(self, input_shape: Tuple[int, int, int], layers: List[int], layer_widths: List[int], latent_dim: int,
                 first_conv: bool = False, maxpool1: bool = False, dropout_prob: float = 0.):
        super().__init__()
        if first_conv or maxpool1:
            raise NotImplementedError("first_conv / maxpool1 stems are not implemented")
        assert len(layers) == len(layer_widths) and len(layers) > 0
        self.input_shape = tuple(input_shape)
        self.latent_dim = latent_dim
        self.conv1, self.bn1 = _conv(input_shape[0], layer_widths[0], 3), nn.BatchNorm2d(layer_widths[0])
        self.layers = nn.ModuleList()
        cin = layer_widths[0]
        for i, (n, width) in enumerate(zip(layers, layer_widths)):
            stride = 1 if i == 0 else 2
            blocks = [_EncBlock(cin, width, stride, stride != 1 or cin != width)]
            blocks += [_EncBlock(width, width, 1, False) for _ in range(1, n)]
            self.layers.append(nn.Sequential(*blocks))
            cin = width
        self.final = nn.Linear(cin, latent_dim)
        self._init_cache()

    @property
    def input_dim(self):
        return prod(self.input_shape)

    @property
    def output_dim(self):
        return self.latent_dim

    def _program(self):
        C, H, W = self.input_shape
        b = _lib.ProgramBuilder(C * H * W, self.latent_dim, self.out_split)
        cur = 1
        h, w = b.conv(0, cur, *_fold(self.conv1, self.bn1), hin=H, win=W, relu=True, in_nchw=C > 1)
        for stage in self.layers:
            for blk in stage:
                t1, t2, t3 = _scratch(cur)
                h2, w2 = b.conv(cur, t1, *_fold(blk.conv1, blk.bn1), hin=h, win=w, stride=blk.stride, relu=True)
                res = cur
                if blk.downsample is not None:
                    b.conv(cur, t2, *_fold(blk.downsample[0], blk.downsample[1]), hin=h, win=w, stride=blk.stride)
                    res = t2
                b.conv(t1, t3, *_fold(blk.conv2, blk.bn2), hin=h2, win=w2, relu=True, res=res)
                cur, h, w = t3, h2, w2
        t1 = _scratch(cur)[0]
        b.avgpool(cur, t1, channels=self.final.in_features, hin=h, win=w)
        b.linear(t1, _lib.CS_NET_BUF_OUT, self.final.weight, self.final.bias)
        return b


class ResNetDecoder(_ProgramNet):
    def __init__(self, output_shape: Tuple[int, int, int], layers: List[int], layer_upscale: List[Optional[int]],
                 layer_widths: List[int], downscaled_resolution: Tuple[int, int], latent_dim: int, upscale_factor: int,
                 first_conv: bool = False, maxpool1: bool = False, dropout_prob: float = 0.):
        super().__init__()
        if first_conv or maxpool1:
            raise NotImplementedError("first_conv / maxpool1 heads are not implemented")
        assert len(layers) == len(layer_widths) and len(layers) > 0
        self.output_shape = tuple(output_shape)
        self.layer_widths = list(layer_widths)
        self.layer_upscale = list(layer_upscale)
        self.downscaled_resolution = tuple(downscaled_resolution)
        self.latent_dim = latent_dim
        self.upscale_factor = upscale_factor
        cin = layer_widths[0] * 2
        self.linear = nn.Linear(latent_dim, cin * prod(downscaled_resolution))
        self.layers = nn.ModuleList()
        for i, (n, width) in enumerate(zip(layers, layer_widths)):
            scale = 2 if i < len(layers) - 1 else 1
            blocks = [_DecBlock(cin, width, scale, scale != 1 or cin != width)]
            blocks += [_DecBlock(width, width, 1, False) for _ in range(1, n)]
            self.layers.append(nn.Sequential(*blocks))
            cin = width
        self.conv1 = _conv(cin, output_shape[0], 3)
        self._init_cache()

    @property
    def input_dim(self):
        return self.linear.in_features

    @property
    def output_dim(self):
        return prod(self.output_shape)

    def _program(self):
        Cout, H, W = self.output_shape
        c = self.layer_widths[0] * 2
        h, w = self.downscaled_resolution
        b = _lib.ProgramBuilder(self.latent_dim, Cout * H * W, self.out_split)
        # the reference views the linear output as [c, h, w]; activations here are [h, w, c]
        perm = torch.arange(c * h * w).view(c, h, w).permute(1, 2, 0).reshape(-1)
        cur = 1
        b.linear(0, cur, self.linear.weight.detach().cpu()[perm], self.linear.bias.detach().cpu()[perm])
        first = H // self.upscale_factor
        if (h, w) != (first, first):
            t1 = _scratch(cur)[0]
            b.resize(cur, t1, channels=c, hin=h, win=w, hout=first, wout=first)
            cur, h, w = t1, first, first
        for i, stage in enumerate(self.layers):
            for blk in stage:
                t1, t2, t3 = _scratch(cur)
                b.conv(cur, t1, *_fold(blk.conv1, blk.bn1), hin=h, win=w, relu=True)
                conv2, scale = _conv_of(blk.conv2)
                res = cur
                if blk.upsample is not None:
                    sc_conv, sc_scale = _conv_of(blk.upsample[0])
                    b.conv(cur, t2, *_fold(sc_conv, blk.upsample[1]), hin=h, win=w, upsample=sc_scale)
                    res = t2
                h, w = b.conv(t1, t3, *_fold(conv2, blk.bn2), hin=h, win=w, upsample=scale, relu=True, res=res)
                cur, c = t3, conv2.out_channels
            up = self.layer_upscale[i] if i < len(self.layer_upscale) else None
            if up is not None and (h, w) != (up, up):
                t1 = _scratch(cur)[0]
                b.resize(cur, t1, channels=c, hin=h, win=w, hout=up, wout=up)
                cur, h, w = t1, up, up
        assert (h, w) == (H, W), f"decoder ends at {h}x{w}, expected {H}x{W}"
        b.conv(cur, _lib.CS_NET_BUF_OUT, self.conv1.weight, None, hin=h, win=w, out_nchw=Cout > 1)
        return b
