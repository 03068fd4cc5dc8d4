"""Deterministic synthetic weights for every sub-network of the VSR path.

The reference ships no checkpoints at all (`.gitignore:1` hides
`pretrained/`, `main.py:121` "Random initialization"), so parity and the
benchmark run on seeded synthetic weights.  The generator is keyed on the
*parameter name* and the *module type*, never on traversal order, and draws
from `numpy.random.RandomState` (a frozen bit-stream), so the very same
numbers are produced

  * in the development container for the reference's own modules (golden
    fixtures: the golden-vector generator), and
  * on the GPU box for this package's modules (which keep the reference's
    `state_dict` key names),

without any weight file ever travelling between the two.
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

_GAIN = math.sqrt(2.0)
# fixed-function layers of the SR net (reference blocks.py:46-55, MeanShift):
# identity weight and +-255*mean bias, never trained, never regenerated.
_SKIP_SUFFIXES = ("sub_mean", "add_mean")
# layers fed with raw 0..255 pixels and followed by no normalisation: scale their weights so the
# synthetic depth / segmentation trunks stay O(1) instead of O(255) (keeps the recurrent
# `estimated_image` loop and half-precision storage well conditioned).  Keyed by name suffix.
_WEIGHT_SCALE = {"netG.0": 1.0 / 255.0, "net.stages.0.0": 1.0 / 128.0}
# gain of the two layers of the fusion MLP (kept < 1 so frame-to-frame recurrence is contractive)
_FC_SCALE = 0.5


def _rs(name: str, seed: int) -> np.random.RandomState:
    return np.random.RandomState((zlib.crc32(name.encode("utf-8")) + 7919 * seed) & 0xFFFFFFFF)


def _normal(name, seed, shape, std):
    return (_rs(name, seed).standard_normal(size=shape) * std).astype(np.float32)


def _uniform(name, seed, shape, lo, hi):
    return _rs(name, seed).uniform(lo, hi, size=shape).astype(np.float32)


def synth_tensors_for_module(root: nn.Module, seed: int = 0, prefix: str = "") -> "OrderedDict[str, np.ndarray]":
    """Return {state_dict key -> float32 array} for every generated entry of `root`.

    Only module *types* and *names* are inspected, so a structurally identical
    module tree (same names, same layer types/shapes) gets identical numbers.
    """
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for mod_name, m in root.named_modules():
        full = prefix + mod_name
        if full.endswith(_SKIP_SUFFIXES):
            continue
        key = (full + ".") if full else ""
        if isinstance(m, nn.ConvTranspose2d):
            kh, kw = m.kernel_size
            sh, sw = m.stride
            fan = m.in_channels * kh * kw / float(sh * sw)
            out[key + "weight"] = _normal(key + "weight", seed, tuple(m.weight.shape), _GAIN / math.sqrt(fan))
            if m.bias is not None:
                out[key + "bias"] = _uniform(key + "bias", seed, tuple(m.bias.shape), -0.05, 0.05)
        elif isinstance(m, nn.Conv2d):
            kh, kw = m.kernel_size
            fan = (m.in_channels // m.groups) * kh * kw
            scale = next((v for k, v in _WEIGHT_SCALE.items() if full.endswith(k)), 1.0)
            out[key + "weight"] = _normal(key + "weight", seed, tuple(m.weight.shape), scale * _GAIN / math.sqrt(fan))
            if m.bias is not None:
                out[key + "bias"] = _uniform(key + "bias", seed, tuple(m.bias.shape), -0.05, 0.05)
        elif isinstance(m, nn.Linear):
            # the 8->32->1 fusion MLP ends in ReLU (reference SRProjectionModule.py:126-131);
            # with a symmetric init its output is ~all zero, so the last
            # layer is kept positive to make the parity target non-degenerate.
            last = m.out_features == 1
            w = _normal(key + "weight", seed, tuple(m.weight.shape), _FC_SCALE / math.sqrt(m.in_features))
            if last:
                w = np.abs(w)
            out[key + "weight"] = w
            if m.bias is not None:
                out[key + "bias"] = _uniform(key + "bias", seed, tuple(m.bias.shape), 0.0 if last else -0.5, 0.5)
        elif isinstance(m, nn.PReLU):
            out[key + "weight"] = _uniform(key + "weight", seed, tuple(m.weight.shape), 0.1, 0.3)
        elif isinstance(m, nn.BatchNorm2d):
            c = m.num_features
            if m.affine:
                out[key + "weight"] = _uniform(key + "weight", seed, (c,), 0.8, 1.2)
                out[key + "bias"] = _normal(key + "bias", seed, (c,), 0.1)
            out[key + "running_mean"] = _normal(key + "running_mean", seed, (c,), 0.1)
            out[key + "running_var"] = _uniform(key + "running_var", seed, (c,), 0.8, 1.2)
    return out


@torch.no_grad()
def fill_module_(root: nn.Module, seed: int = 0, prefix: str = "") -> nn.Module:
    """Overwrite `root`'s parameters/buffers in place with the synthetic set."""
    tensors = synth_tensors_for_module(root, seed, prefix)
    sd = root.state_dict()
    for k, v in tensors.items():
        local = k[len(prefix):]
        if local not in sd:
            raise KeyError(f"synthetic weight {k!r} has no state_dict entry")
        if tuple(sd[local].shape) != v.shape:
            raise ValueError(f"{k}: shape {tuple(sd[local].shape)} vs generated {v.shape}")
        sd[local].copy_(torch.from_numpy(v))
    return root
