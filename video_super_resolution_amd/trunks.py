"""Execution copies of the stock-convolution guidance trunks.

The trunks (FlowNet2, the depth hourglass, OSVOS) keep float32 master parameters under the reference's state_dict
keys.  For the throughput configuration a derived copy is executed instead: eval-mode BatchNorm folded into the
preceding convolution (the hourglass has 155 of them: one elementwise kernel and one read+write of the activation
each) and parameters cast to float16.  Copies are rebuilt when any master parameter changes.
"""
from __future__ import annotations

import copy

import torch
import torch.nn as nn


def _fold_bn_(seq: nn.Sequential) -> None:
    """In a Sequential, fold every `Conv2d -> BatchNorm2d` pair (eval statistics) into the conv; recurse."""
    mods = list(seq._modules.items())
    for idx, (name, m) in enumerate(mods):
        if isinstance(m, nn.Sequential):
            _fold_bn_(m)
        elif isinstance(m, nn.BatchNorm2d) and idx > 0 and isinstance(mods[idx - 1][1], nn.Conv2d):
            conv = mods[idx - 1][1]
            scale = torch.rsqrt(m.running_var + m.eps)
            if m.affine:
                scale = scale * m.weight
            shift = -m.running_mean * scale
            if m.affine:
                shift = shift + m.bias
            conv.weight.data.mul_(scale.view(-1, 1, 1, 1))
            if conv.bias is None:
                conv.bias = nn.Parameter(torch.zeros_like(shift))
            conv.bias.data.mul_(scale).add_(shift)
            seq._modules[name] = nn.Identity()


def _walk_fold(module: nn.Module) -> None:
    for child in module.children():
        _walk_fold(child)
    if isinstance(module, nn.Sequential):
        _fold_bn_(module)


class ExecCopy:
    """Lazily built, version-checked derived copy of a trunk."""

    def __init__(self, master: nn.Module, fold_bn: bool):
        self.master = master
        self.fold_bn = fold_bn
        self._copy = None
        self._key = None

    def _version(self, dtype):
        return (dtype,) + tuple((t.data_ptr(), t._version) for t in list(self.master.parameters()) + list(self.master.buffers()))

    @torch.no_grad()
    def get(self, dtype: torch.dtype) -> nn.Module:
        if dtype == torch.float32 and not self.fold_bn:
            return self.master
        key = self._version(dtype)
        if self._copy is None or key != self._key:
            c = copy.deepcopy(self.master).eval()
            if self.fold_bn:
                _walk_fold(c)
            self._copy = c.to(dtype)
            self._key = key
        return self._copy
