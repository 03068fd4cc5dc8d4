"""Inference calls of `VSR.forward` replayed as ONE HIP graph.

A forward is ~900 kernel launches on four streams; issued eagerly the host needs ~11 ms of a 23 ms frame for them (LAB_NOTES R5.9),
which a slow or contended host turns into the bound.  `GraphedVSR` records the launches of one call -- the side streams fork from and
join the capturing stream through the events `VSR.forward` already uses -- and replays them per frame: one `hipGraphLaunch` instead
of ~900 Python -> ctypes -> HIP calls, same kernels, same values (tests/test_gpu_vsr.py::test_graph_replay_is_bit_identical_to_eager).

Opt-in, inference only (`train=False`, no `target`): the reference's calling convention is kept (video_super_resolution.py:23,66-69),
`high_frames[1] = out` is done after the replay.  A graph is bound to the input geometry, to whether a previous output is given, to the
module's switches and to the (address, version) of every parameter and buffer at capture time -- the kernels read PACKED copies of the
weights, so after `load_state_dict` / an optimizer step / `.half()` the next call captures afresh instead of replaying stale weights.
"""
from __future__ import annotations

import torch

from .vsr import VSR


class GraphedVSR:
    def __init__(self, model: VSR, clone_output: bool = True):
        """`clone_output=False` returns the graph's own output buffer (overwritten by the next call of the same geometry)."""
        self.model = model
        self.clone_output = clone_output
        self._graphs = {}

    def reset(self):
        self._graphs = {}

    def _key(self, data, est):
        m = self.model
        weights = (m.model._weights_key(), m._flow_exec.key(), m._depth_exec.key(), m._vos_exec.key())
        return (tuple(data.shape), data.dtype, data.device.index, None if est is None else (tuple(est.shape), est.dtype),
                m.precision, m.upscale_factor, m.share_planes, m.share_tail, m.overlap_shared, m.early_planes, m.depth_est_late, weights)

    def _capture(self, data, est):
        m = self.model
        sd = data.detach().clone()
        se = None if est is None else est.detach().clone()
        # one eager call on a side stream first: lazy state (packed weights, executors, streams, LDS attributes) is built outside the capture
        s = torch.cuda.Stream(device=data.device)
        s.wait_stream(torch.cuda.current_stream(data.device))
        with torch.cuda.stream(s), torch.no_grad():
            m(sd, None, None, se, train=False)
        torch.cuda.current_stream(data.device).wait_stream(s)
        torch.cuda.synchronize(data.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g), torch.no_grad():
            so, _ = m(sd, None, None, se, train=False)
        return g, sd, se, so

    def __call__(self, data, target, high_frames, estimated_image, train=False):
        m = self.model
        if train or target is not None or torch.is_grad_enabled() and m.training:
            raise ValueError("GraphedVSR replays inference calls only (train=False, target=None, under no_grad or in eval mode): "
                             "call the module itself for a training step")
        if m.temporal_cache:
            raise ValueError("VSR.temporal_cache keys on the identity of the frame tensors: not usable with the static buffers of a graph")
        if not data.is_cuda:
            raise RuntimeError("VSR runs on the GPU through hand-written HIP kernels; there is no CPU fallback")
        with torch.cuda.device(data.device):
            key = self._key(data, estimated_image)
            ent = self._graphs.get(key)
            if ent is None:
                if len(self._graphs) >= 8:     # (each graph owns its intermediates: a service that sees many geometries / weight versions keeps the latest)
                    self._graphs.pop(next(iter(self._graphs)))
                ent = self._graphs[key] = self._capture(data, estimated_image)
            g, sd, se, so = ent
            sd.copy_(data)
            if se is not None:
                se.copy_(estimated_image)
            g.replay()
            out = so.clone() if self.clone_output else so
            if high_frames is not None:
                high_frames[1] = out.detach()   # video_super_resolution.py:66
        return out, None
