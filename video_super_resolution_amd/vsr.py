"""`VSR`: the per-frame video-SR forward of the reference behind its own `nn.Module` API.

Drop-in for `network/video_super_resolution.py:12-69`:

    VSR()                                            no-arg constructor, is an nn.Module       (:13-21)
    forward(data, target, high_frames, estimated_image, train=True) -> (output, loss)            (:23-69)
        data            [3,h,w,3]  float 0..255 LR frames (NHWC)
        high_frames     [3,4h,4w,3]; slot 1 is overwritten with the output in place            (:66)
        estimated_image None | [1,4h,4w,3] (the previous output, fed back verbatim, main.py:199-203)
        output          [1,4h,4w,3] float32 on the module's device
    sub-module names .model .FlowModule .DepthModule .VOSModule are kept; `.model.state_dict()` has the
    reference's keys (main.py:118,235).

What differs, on purpose:
  * everything stays on the device: the four host round trips of the flow colour coding, the two of the
    VOS wrapper and the numpy masked-array fill (:58-60) are device kernels / device tensor ops;
  * the depth trunk is evaluated once per distinct frame (the reference evaluates frame 1 of every
    triplet twice and frame 2 again in the second pass: 8 trunk runs, 4-5 distinct inputs);
  * the loss branch (`train=True`, :67,:71-80) is evaluated by loss.py (the reference's SR_loss / Flow_loss /
    GetObjectsForOBJLoss with its own names) under no_grad on the device and returns the reference's 0-d CPU tensor, so the
    reference driver's call `model(x, y, high_frame, estimated_image)` + `real_loss.data` (main.py:199-203) works as is.
    Inference (`train=False`) returns `loss=None`;
  * the train step's single differentiable call (:64 at main.py:205-210) is served by `sr_train.forward_train` (forward and
    backward of the SR net on this repository's own float32 kernels, csrc/sr_train.hip) when the module is in training mode
    with autograd enabled; eval mode or `no_grad` runs the inference kernels.  (`SRProjectionModule._forward_autograd`, the
    stock-operator restatement, is a cross-check for the tests only.)
"""
from __future__ import annotations

import os

from typing import Callable, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .depth import DepthProjectionModule
from .loss import Flow_loss, GetObjectsForOBJLoss, SR_loss, loss_calculate
from .flownet import FlowProjectionModule
from .sr import SRProjectionModule
from .trunk_exec import FlowNet2Exec, HourglassExec, OSVOSExec, TrunkExecCache
from .vos import VOSProjectionModule
from ._lib import on_device as _on_device


def maskprocess(mask: torch.Tensor) -> torch.Tensor:
    """utils/tools.py:76-77: replicate a [h,w] map to three channels."""
    return torch.stack((mask,) * 3)


class VSR(nn.Module):
    def __init__(self, upscale_factor: int = 4):
        """`VSR()` is the reference's constructor (x4, video_super_resolution.py:13-21).  `upscale_factor` 2 / 3 select the
        scale extension of the SR net (sr.sr_geometry; BASELINE configs C1, C2, C3-B, C5) -- every other part of the path
        is scale-free."""
        super().__init__()
        self.model = SRProjectionModule(upscale_factor=upscale_factor)
        self.upscale_factor = upscale_factor
        self.FlowModule = FlowProjectionModule().eval()
        self.DepthModule = DepthProjectionModule().eval()
        self.VOSModule = VOSProjectionModule().eval()
        # the train=True branch (video_super_resolution.py:19-21,67,71-80): same sub-module names as the reference
        self.SR_loss = SR_loss().eval()
        self.Flow_loss = Flow_loss().eval()
        self.loss4object = GetObjectsForOBJLoss().eval()
        self.loss_fn: Optional[Callable] = None   # optional override: loss_fn(target, high_frames) instead of loss_calculate
        # "fp16": the headline configuration -- SR stack on the MFMA path, guidance trunks on the hand-written NHWC fp16
        #         MFMA convolution (trunk_exec.py: BatchNorm folded, concatenations written in place, frames batched);
        # "fp32": SR stack in exact float32 kernels, trunks on stock float32 convolutions (the parity configuration).
        self.precision = "fp16"
        self.share_planes = True   # evaluate the three LR-frame planes once per forward (both SR passes read them)
        self.share_tail = os.environ.get("VSR_SHARE_TAIL", "1") != "0"   # ... their tail (pre-fusion planes) too (A/B switch)
        self.f32_streams = os.environ.get("VSR_F32_STREAMS", "1") != "0"   # float32 configuration: the trunks (and the shared planes' SR maps) on separate streams too (A/B switch; +7.6 % same box)
        self.overlap_shared = True  # ... and do so on a side stream beside the guidance trunks of pass 1 (fp16 configuration)
        # ... and the planes of a pass that are known BEFORE its guidance trunks finish: plane 7 of pass 1 (the resized previous output,
        # :37-38) beside the pass-1 trunks, plane 7 of pass 2 (the masked pass-1 frame, :58-60) behind OSVOS on its stream, beside FlowNet2
        # (SRProjectionModule.precompute_rows; same kernels on the same values: bit-identical frames)
        # level 2: also the two depth planes of pass 2 behind the hourglass on its stream (beside FlowNet2, the long pole of pass 2's guidance);
        # level 3: also the two flow-picture planes of pass 1 on the main stream behind FlowNet2 (beside the hourglass, pass 1's long pole)
        self.early_planes = int(os.environ.get("VSR_EARLY_PLANES", "1"))
        # the hourglass on the estimate (only pass 2 reads its depth): True = with pass 2's guidance, as a batch of two with the pass-1 frame,
        # in the shadow of FlowNet2 (pass 2's long pole); False = batched with the three LR frames in pass 1 (where the hourglass IS the long pole)
        self.depth_est_late = os.environ.get("VSR_DEPTH_EST_LATE", "0") != "0"   # (measured level: guidance 1 -1.0 ms, guidance 2 +0.9 ms, profiles/r05_early_planes_stages.txt: off)
        self.early_scales = (4,)   # x2 (C3-B): measured level with the plain order (14.15 / 14.26 vs 14.17 / 14.16 frames/s, same box): off there
        # Opt-in streaming mode (OFF by default; the headline benchmark leaves it off): consecutive windows of a clip share
        # two of their three LR frames (utils/video_utils.py:25), so the depth prediction of a frame and the flow picture of a
        # frame pair computed for window t are what window t+1 computes again.  With temporal_cache = True they are kept
        # across calls, keyed by the identity AND version counter of the frame tensors (views of one clip tensor, as
        # main.py:196-199 / driver.run_item hand them over): about two of G1's four hourglass runs and one of its two FlowNet2
        # runs saved per window.  Same networks on the same frames, but on smaller batches -- the MFMA convolution picks its
        # tile / split-K shape by the pixel count -- so the frames agree with the per-window evaluation to rounding, not bit for bit.
        # A cache entry HOLDS the frame tensors it was computed from, so their storage cannot be freed and handed out again
        # under the same address (windows built afresh by torch.stack / .to(dev) never hit: different storage, and the old
        # one is still alive); what the key cannot see is a write into the SAME storage that bypasses the version counter
        # (a raw-pointer kernel launch): call `reset_temporal_cache()` after one.
        self.temporal_cache = False
        self._tcache = {"depth": {}, "flow": {}}
        self.keep_loss_terms = False   # True: `last_loss_terms` = the four terms + masked tensors of the last train=True call
        self.last_loss_terms = None
        self._flow_exec = TrunkExecCache(self.FlowModule.net, FlowNet2Exec)
        self._depth_exec = TrunkExecCache(self.DepthModule.model.netG, HourglassExec)
        self._vos_exec = TrunkExecCache(self.VOSModule.net, OSVOSExec)

    def train(self, mode: bool = True):
        # main.py:178 calls model.train(), which would flip the frozen guidance networks (HG BatchNorm!)
        # out of eval mode; the reference's constructor intends them frozen (:16-21).  Keep them in eval.
        super().train(mode)
        for m in (self.FlowModule, self.DepthModule, self.VOSModule, self.SR_loss, self.Flow_loss, self.loss4object):
            m.eval()
        return self

    def loss_calculate(self, target, outputs):
        """video_super_resolution.py:71-80 (a 0-d CPU tensor, computed under no_grad like the reference)."""
        taps = {} if self.keep_loss_terms else None
        loss = loss_calculate(self, target, outputs, taps)
        if taps is not None:
            self.last_loss_terms = taps
        return loss

    def __deepcopy__(self, memo):
        """copy.deepcopy of a module that has run: HIP streams and the executors built on them are per-instance run-time
        state (not picklable, and not meant to be shared): the copy gets fresh ones."""
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        runtime = {"_streams", "_streams_key", "_flow_exec", "_depth_exec", "_vos_exec", "_tcache"}
        for k, v in self.__dict__.items():
            if k not in runtime:
                new.__dict__[k] = copy.deepcopy(v, memo)
        new._flow_exec = TrunkExecCache(new.FlowModule.net, FlowNet2Exec)
        new._depth_exec = TrunkExecCache(new.DepthModule.model.netG, HourglassExec)
        new._vos_exec = TrunkExecCache(new.VOSModule.net, OSVOSExec)
        new._tcache = {"depth": {}, "flow": {}}
        return new

    def reset_temporal_cache(self):
        self._tcache = {"depth": {}, "flow": {}}

    @staticmethod
    def _tkey(t):
        return (t.data_ptr(), tuple(t.shape), t._version, t.device.index)

    # ------------------------------------------------------------------------------------------
    def _fast(self) -> bool:
        if self.precision not in ("fp16", "fp32"):
            raise ValueError(f"precision must be 'fp16' or 'fp32', got {self.precision!r}")
        return self.precision == "fp16"

    def _side_streams(self, dev):
        key = (dev.type, dev.index)
        if getattr(self, "_streams_key", None) != key:
            # [0]: depth trunk; [1]: VOS trunk (pass 2 only) AND the shared planes' SR maps (pass 1 only) -- they never overlap, and with the main
            # stream and FlowNetSD's this makes four streams: HIP's four hardware queues (a fifth ACTIVE queue costs 20 %: LAB_NOTES R4.7)
            self._streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
            self._streams_key = key
        return self._streams

    @torch.no_grad()
    def _guidance(self, trip, depth_cache, extra_depth=(), with_vos=None, cacheable=False, after_vos=None, after_depth=None, after_flow=None):
        """trip: three [h,w,3] frames -> (flow pictures [2,h',w',3], the three single-frame depth predictions [1,1,h,w]
        [, VOS mask [h,w]]): what `_assemble` turns into planes 3-6 (and the masked plane 7) of the SR input.

        The flow, depth and segmentation trunks are independent of each other and individually too small to fill 256
        CUs in their low-resolution layers, so in the fp16 configuration they run concurrently on separate HIP streams
        (joined before the 8-plane assembly).  `extra_depth`: frames whose depth is wanted later (batched now)."""
        h, w = trip[0].shape[:2]
        fast = self._fast()
        par = fast or self.f32_streams     # the trunks on their own streams (float32 configuration: opt-in, see __init__)
        main = torch.cuda.current_stream(trip[0].device)
        s_depth, s_vos = self._side_streams(trip[0].device)[:2] if par else (main, main)
        if par:
            s_depth.wait_stream(main)
            s_vos.wait_stream(main)

        # depth trunk once per distinct frame, all new frames as one batch
        tc = self._tcache if (self.temporal_cache and cacheable) else None
        if tc is not None:   # streaming mode: predictions of frames an earlier window already saw
            for f in trip:
                hit = tc["depth"].get(self._tkey(f))
                if hit is not None and f.data_ptr() not in depth_cache:
                    depth_cache[f.data_ptr()] = (f, hit[1])
        new = []
        for f in list(trip) + list(extra_depth):
            if f.data_ptr() not in depth_cache and all(f.data_ptr() != g.data_ptr() for g in new):
                new.append(f)
        if new:
            # pass 1's hourglass batch is "the window's frames + the estimate" (4); fewer run when the estimate IS frame 0 (first call) or,
            # in streaming mode, when predictions of shared frames are cached -- the launchers still choose the kernels of the batch of
            # 4, so a frame's prediction does not depend on the batch it travelled in (streaming == per-window evaluation, bit for bit)
            n_window = (len(trip) + len(extra_depth)) if cacheable else 0
            from . import _lib as L
            with torch.cuda.stream(s_depth), L.route_batch(n_window, len(new) if n_window != len(new) else 0):
                if fast:
                    z = self._depth_exec.get()(torch.stack(new))  # [k,1,h,w] float32
                else:
                    z = self.DepthModule.model(torch.stack(new).permute(0, 3, 1, 2))
                z.record_stream(main)
            for i, f in enumerate(new):
                depth_cache[f.data_ptr()] = (f, z[i:i + 1])  # keep f alive so the pointer stays unique
        if after_depth is not None:
            with torch.cuda.stream(s_depth):
                after_depth([depth_cache[f.data_ptr()][1] for f in trip])   # (more work for this stream, behind the predictions)
        mask = None
        if with_vos is not None:
            with torch.cuda.stream(s_vos):
                mask = self.VOSModule(with_vos[0], with_vos[1], self._vos_exec.get() if fast else None)  # [h,w] in {0,1}
                mask.record_stream(main)
                if after_vos is not None:
                    after_vos(mask)   # (more work for this stream, behind the mask)
        # both frame pairs as one FlowNet2 batch of two (on the main stream)
        pairs = [(trip[0], trip[1]), (trip[1], trip[2])]
        net = self._flow_exec.get() if fast else None
        if tc is not None:
            keys = [(self._tkey(a), self._tkey(b)) for a, b in pairs]
            have = [tc["flow"].get(k) for k in keys]
            have = [hv[2] if hv is not None else None for hv in have]
            todo = [p for p, hv in zip(pairs, have) if hv is None]
            from . import _lib as L
            with L.route_batch(len(pairs), len(todo) if 0 < len(todo) < len(pairs) else 0):   # (a cached pair is missing from the batch: see above)
                fresh = iter(self.FlowModule.forward_pairs(todo, net)) if todo else iter(())
            pics_l = [hv if hv is not None else next(fresh) for hv in have]
            pics = torch.stack(pics_l)
            # keep this window's two pictures only -- WITH their frames, so the keyed addresses stay taken
            tc["flow"] = {k: (a, b, p) for k, (a, b), p in zip(keys, pairs, pics_l)}
        else:
            pics = self.FlowModule.forward_pairs(pairs, net)
        if after_flow is not None:
            after_flow(pics)   # (more work for the main stream, behind the flow pictures, while the other trunks finish)
        if par:
            main.wait_stream(s_depth)
            main.wait_stream(s_vos)
        z = [depth_cache[f.data_ptr()][1] for f in trip]
        if tc is not None:
            tc["depth"] = {self._tkey(f): (f, zz) for f, zz in zip(trip, z)}        # ... and its three depth predictions
        return pics, z, mask

    @staticmethod
    def _assemble(frames_nhwc, pics, z, est_chw=None, mask=None):
        """The 8-plane SR input of :33-40 / :57-62 in one launch (csrc/flow_ops.hip k_assemble_planes): frames NHWC -> NCHW,
        flow pictures resized to h x w (nearest, :35,:52), depth = mean of two predictions x3 (DepthProjectionModule.py:16,
        tools.py:76-77), estimate plane (frame 0 | given plane), zeroed under the VOS mask (:58-60)."""
        from . import _lib as L
        _, h, w, _ = frames_nhwc.shape
        out = torch.empty((8, 3, h, w), dtype=torch.float32, device=frames_nhwc.device)
        zc = [t.reshape(h, w).contiguous() for t in z]
        L.check(L.load().vsr_assemble_planes_f32(L.dptr(frames_nhwc), L.dptr(pics), pics.shape[1], pics.shape[2], L.dptr(zc[0]),
                                                 L.dptr(zc[1]), L.dptr(zc[2]), L.optr(est_chw), L.optr(mask), L.dptr(out), h, w,
                                                 L.stream()), "assemble_planes")
        return out

    @_on_device
    def forward(self, data, target, high_frames, estimated_image, train=True):
        if data.dim() != 4 or data.shape[0] != 3 or data.shape[3] != 3:
            raise ValueError(f"data must be [3,h,w,3], got {tuple(data.shape)}")
        if not data.is_cuda:
            raise RuntimeError("VSR runs on the GPU through hand-written HIP kernels; there is no CPU fallback "
                               "(move the module and its inputs to the device first)")
        ev = self.stage_events = [] if getattr(self, "stage_timing", False) else None

        def mark():   # (measurement hook, tools/frame_stages.py: events on the main stream at the four stage boundaries)
            if ev is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                ev.append(e)
        with torch.no_grad():
            h, w = data.shape[1], data.shape[2]
            d = data.detach().to(torch.float32).contiguous()
            f0, f1, f2 = d[0], d[1], d[2]
            depth_cache = {}
            mark()

            # ---- pass 1 (:26-41)
            if estimated_image is None:
                est, est_hw3 = None, f0   # plane 7 = frame 0 (:38)
            else:                          # :37 nearest resize of the previous output, as a plane and as a frame
                prev = estimated_image.detach().to(torch.float32).contiguous()
                est = torch.empty((3, h, w), dtype=torch.float32, device=d.device)
                est_hw3 = torch.empty((h, w, 3), dtype=torch.float32, device=d.device)
                from . import _lib as L
                L.check(L.load().vsr_resize_estimate_f32(L.dptr(prev), prev.shape[1], prev.shape[2], L.dptr(est), L.dptr(est_hw3), h, w,
                                                         L.stream()), "resize_estimate")
            self.model.precision = self.precision
            # planes 0-2 of both SR calls are the LR frames themselves (:33, :57: `data` NHWC -> NCHW): their FeedbackBlock maps
            # depend on nothing the guidance trunks produce, so they are evaluated NOW, on a side stream, next to those trunks
            # (whose low-resolution layers leave most CUs idle); both SR calls then run head + FeedbackBlock on planes 3-7 only
            shared = {"n": 3} if self.share_planes else None
            s_sr = None
            if shared is not None and self.overlap_shared and not self._fast() and self.f32_streams:
                # float32 configuration: the same overlap; the SR module keeps the planes' pre-fusion maps in `shared`
                main = torch.cuda.current_stream(d.device)
                x_first = d.permute(0, 3, 1, 2).contiguous()
                s_sr = self._side_streams(d.device)[1]
                s_sr.wait_stream(main)
                with torch.cuda.stream(s_sr):
                    self.model.precompute_shared(x_first, shared, None)
                    shared["prefc_f32"].record_stream(main)
                x_first.record_stream(s_sr)
            elif shared is not None and self.overlap_shared and self._fast() and self.model.block.num_groups == 6:
                main = torch.cuda.current_stream(d.device)
                x_first = d.permute(0, 3, 1, 2).contiguous()
                n_planes = self.model.fc[0].in_features   # 8: the planes of one SR call (video_super_resolution.py:40)
                live = {k: torch.empty((n_planes, h * w, 32), dtype=torch.float16, device=d.device) for k in (3, 6)}
                if self.model.upscale_factor in (4, 2) and self.share_tail:
                    # ... and their pre-fusion planes (the tail's output) at full resolution: both passes' tails skip them
                    S = self.model.upscale_factor
                    live["prefc"] = torch.empty((n_planes, 3, S * h, S * w), dtype=torch.float32, device=d.device)
                s_sr = self._side_streams(d.device)[1]
                s_sr.wait_stream(main)
                early = self.early_planes if self.model.upscale_factor in self.early_scales else 0
                with torch.cuda.stream(s_sr):
                    self.model.precompute_shared(x_first, shared, live)
                    if early:   # plane 7 of pass 1: the previous output at h x w (:37), frame 0 on the first call (:38)
                        self.model.precompute_rows((est if est is not None else x_first[0]).unsqueeze(0), live, n_planes - 1)
                        shared["todo"] = (3, n_planes - 1)
                x_first.record_stream(s_sr)
                if est is not None:
                    est.record_stream(s_sr)
                for t in live.values():
                    t.record_stream(s_sr)
            # (the estimate's depth is only used in pass 2 but is already known: batched with the three frames)
            after_flow = None
            if shared is not None and shared.get("todo") is not None and self.early_planes >= 3:
                def after_flow(p):   # planes 3, 4 of pass 1: the flow pictures at h x w (:35), as k_assemble_planes writes them
                    zd = torch.zeros((1, 1, h, w), dtype=torch.float32, device=d.device)
                    self.model.precompute_rows(self._assemble(d, p, [zd, zd, zd], est)[3:5], shared["live"], 3)
                    shared["todo"] = (5, shared["todo"][1])
            pics, z, _ = self._guidance((f0, f1, f2), depth_cache, extra_depth=() if (self.depth_est_late and self._fast()) else (est_hw3,), cacheable=True,
                                        after_flow=after_flow)
            if s_sr is not None:
                torch.cuda.current_stream(d.device).wait_stream(s_sr)
            mark()
            # pass 1's frame is only ever read through the nearest x1/4 resize of :44, i.e. at its pixels (4i,4j): the SR
            # stack evaluates its tail and fusion MLP at exactly those (identical values, 1/16 of the tail work)
            # planes 0-2 (the LR frames) are the same in both SR calls (:40, :62): their FeedbackBlock maps are computed here
            # and kept for pass 2 (sr.py:_forward_f16 `shared`; identical values, 3/8 of pass 2's trunk not recomputed)
            x1 = self._assemble(d, pics, z, est)
            mid = self.model(x1, decimate=True, shared=shared)[0]  # [3,h,w] = F.interpolate(out1,(h,w))[0]

            # ---- pass 2 guidance on (estimate, x4-decimated pass-1 output, frame 2) (:43-54)
            mark()
            mid_hw3 = mid.permute(1, 2, 0).contiguous()
            after_vos = after_depth = None
            if shared is not None and shared.get("todo") is not None:
                s_depth, s_vos = self._side_streams(d.device)[:2]
                n_pl = self.model.fc[0].in_features
                shared["todo"] = (3, n_pl - 1)
                if self.early_planes >= 2:
                    def after_depth(z3):   # planes 5, 6 of pass 2: the depth planes (:49-50), as k_assemble_planes writes them
                        self.model.precompute_rows(self._assemble(d, pics, z3, mid)[5:7], shared["live"], 5)
                    shared["todo"] = (3, 5)
                    pics.record_stream(s_depth)
                    mid.record_stream(s_depth)
                    for t in shared["live"].values():
                        t.record_stream(s_depth)

                def after_vos(m):   # plane 7 of pass 2 = the pass-1 frame, zero under the mask (:58-60; as k_assemble_planes writes it)
                    x7 = torch.where(m != 0, torch.zeros_like(mid), mid).unsqueeze(0)
                    self.model.precompute_rows(x7, shared["live"], self.model.fc[0].in_features - 1)
                mid.record_stream(s_vos)
                for t in shared["live"].values():
                    t.record_stream(s_vos)
            pics2, z2, mask = self._guidance((est_hw3, mid_hw3, f2), depth_cache, with_vos=(est_hw3, mid_hw3), after_vos=after_vos, after_depth=after_depth)
            mark()
            x8 = self._assemble(d, pics2, z2, mid.contiguous(), mask.contiguous())   # plane 7: mid, zero where mask != 0 (:58-60)
            if getattr(self, "plane_taps", None) is not None:   # (measurement hook: the SR inputs of both passes and the mask, bench.py / tests)
                self.plane_taps.update(pass1_input=x1.clone(), pass2_input=x8.clone(), vos_mask=mask.clone(), pass1_decimated=mid.clone())
        # ---- pass 2 SR (:62-64): the reference's only call outside no_grad.  Under the caller's no_grad or in eval mode it
        # runs the kernels; in training mode with autograd on it is differentiable (SRProjectionModule.forward)
        out = self.model(x8, shared=shared).permute(0, 2, 3, 1)
        mark()
        with torch.no_grad():
            if high_frames is not None:
                high_frames[1] = out.detach()  # :66
        loss = None
        if train:   # :67 `loss = self.loss_calculate(target, high_frames) if train else None`
            if high_frames is None or target is None:
                raise ValueError("train=True needs `target` [1,H,W,3] and `high_frames` [3,H,W,3] (video_super_resolution.py:67,71-80)")
            loss = self.loss_fn(target, high_frames) if self.loss_fn is not None else self.loss_calculate(target, high_frames)
        return out, loss
