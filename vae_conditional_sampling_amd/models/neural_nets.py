"""Residual MLP with the reference's constructor and state-dict layout, computed by the fused
HIP kernel.

Mirrors ``ResidualFCNetwork`` (irwg/models/neural_nets.py:106-169): parameters are registered
under the same names (``initial_layer``, ``blocks.{i}.linear_layers.{0,1}``, ``final_layer``)
so that the ``generator_network.*`` / ``var_latent_network.*`` entries of a Lightning
checkpoint load unchanged.  ``forward`` packs the weights into the MFMA operand stream on
first use (re-packed after ``load_state_dict`` / ``.to``) and calls ``cs_mlp_forward``.
"""
from typing import Callable, Union

import torch
import torch.nn as nn

from .. import _lib


class _Block(nn.Module):
    """Parameters of one residual block under the reference's names (neural_nets.py:76-85)."""

    def __init__(self, features, use_batch_norm=False, use_layer_norm=False):
        super().__init__()
        assert not (use_batch_norm and use_layer_norm), 'Cannot use both batch and layer normalization.'
        self.linear_layers = nn.ModuleList([nn.Linear(features, features) for _ in range(2)])
        self.use_batch_norm, self.use_layer_norm = use_batch_norm, use_layer_norm
        if use_batch_norm:
            self.batch_norm = nn.BatchNorm1d(features)
        if use_layer_norm:
            self.layer_norm = nn.LayerNorm(features)

    def first_linear(self):
        """(weight, bias) of the first linear layer with an eval-mode BatchNorm folded in
        (float64 arithmetic, rounded once): bn(W h + b) = (s*W) h + (b - mean)*s + beta."""
        W, b = self.linear_layers[0].weight, self.linear_layers[0].bias
        if not self.use_batch_norm:
            return W, b
        bn = self.batch_norm
        s = bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)
        return ((W.double() * s[:, None]).float(), ((b.double() - bn.running_mean.double()) * s + bn.bias.double()).float())


class ResidualFCNetwork(nn.Module):
    def __init__(self, input_dim: int, output_dim: int, num_residual_blocks: int, residual_block_dim: int, *,
                 activation: Union[Callable, str] = "relu", dropout_probability: float = 0.0,
                 use_batch_norm: bool = False, use_layer_norm: bool = False):
        super().__init__()
        if not (activation == "relu" or getattr(activation, "__name__", "") == "relu"):
            raise NotImplementedError("only the ReLU activation is implemented")
        self.input_dims = input_dim
        self.output_dim = output_dim
        self.num_residual_blocks = num_residual_blocks
        self.residual_block_dim = residual_block_dim
        self.dropout_probability = dropout_probability  # identity in eval mode
        self.initial_layer = nn.Linear(input_dim, residual_block_dim)
        self.use_layer_norm = use_layer_norm
        self.blocks = nn.ModuleList([_Block(residual_block_dim, use_batch_norm, use_layer_norm)
                                     for _ in range(num_residual_blocks)])
        self.final_layer = nn.Linear(residual_block_dim, output_dim)
        self.out_split = 2  # (loc | raw scale); the VAE sets 1 for Bernoulli logits, 3 for Student-t
        self.force_program = False  # True: run as a layer program even where the fused kernel would fit
        self._packed = None
        self._packed_key = None
        self._tensors = None
        self.requires_grad_(False)

    @property
    def input_dim(self):
        return self.input_dims

    def weight_dict(self):
        blocks = [b.first_linear() + (b.linear_layers[1].weight, b.linear_layers[1].bias) for b in self.blocks]
        return {"W0": self.initial_layer.weight, "b0": self.initial_layer.bias, "blocks": blocks,
                "Wf": self.final_layer.weight, "bf": self.final_layer.bias}

    def load_weight_dict(self, w):
        with torch.no_grad():
            self.initial_layer.weight.copy_(w["W0"]); self.initial_layer.bias.copy_(w["b0"])
            for blk, (W1, b1, W2, b2) in zip(self.blocks, w["blocks"]):
                blk.linear_layers[0].weight.copy_(W1); blk.linear_layers[0].bias.copy_(b1)
                blk.linear_layers[1].weight.copy_(W2); blk.linear_layers[1].bias.copy_(b2)
            self.final_layer.weight.copy_(w["Wf"]); self.final_layer.bias.copy_(w["bf"])
        self._packed = None
        return self

    def packed(self):
        """Device handle of the packed network (built lazily, rebuilt when weights change)."""
        # called every sampler iteration: the tensor list is gathered once (module traversal is the
        # expensive part), storage pointers and version counters detect .to() / load_state_dict / in-place edits
        if self._tensors is None:
            self._tensors = list(self.parameters()) + list(self.buffers())
        key = (tuple((p.data_ptr(), p._version) for p in self._tensors), self.out_split, torch.cuda.current_device(),
               self.force_program)
        if self._packed is None or self._packed_key != key:
            w = self.weight_dict()
            half = self.output_dim // self.out_split
            if (self.residual_block_dim in (64, 128, 256) and self.input_dims <= 128 and not self.use_layer_norm
                    and self.out_split in (1, 2) and not self.force_program
                    and half <= (64 if self.out_split == 2 else 128)):
                self._packed = _lib.PackedMLP(w["W0"], w["b0"], w["blocks"], w["Wf"], w["bf"], self.out_split)
            else:
                self._packed = self._layer_program(w)
            self._packed_key = key
        return self._packed

    def _layer_program(self, w):
        """Widths outside the fused kernel's range (e.g. the toy configuration's 512) and blocks with
        a LayerNorm: one matrix-core GEMM launch per layer, bias / residual / ReLU in its epilogue,
        the LayerNorm (+ ReLU) as a row-wise launch between the block's two GEMMs."""
        b = _lib.ProgramBuilder(self.input_dims, self.output_dim, self.out_split)
        cur, t, nxt, u = 1, 2, 3, 4
        b.linear(0, cur, w["W0"], w["b0"], relu=True)
        for blk, (W1, b1, W2, b2) in zip(self.blocks, w["blocks"]):
            if blk.use_layer_norm:
                b.linear(cur, u, W1, b1)
                b.layernorm(u, t, blk.layer_norm.weight, blk.layer_norm.bias, relu=True)
            else:
                b.linear(cur, t, W1, b1, relu=True)
            b.linear(t, nxt, W2, b2, relu=True, res=cur)
            cur, nxt = nxt, cur
        b.linear(cur, _lib.CS_NET_BUF_OUT, w["Wf"], w["bf"])
        return b.build()

    def forward(self, inputs: torch.Tensor):
        if self.training:
            raise NotImplementedError("training mode (dropout) is not implemented; call .eval()")
        return self.packed().forward(inputs)
