"""A `main.py`-like driver around `VSR.forward`: the callers and data formats either side of the path.

What the reference's train/validate loop does per dataset item (main.py:154-203) and what utils/video_utils.py:7-33
feeds it, on the GPU path and without OpenCV:

    VideoDataset            sliding 3-frame windows over a decoded RGB clip, cut into `splitvideonum` = 20 chunks per video,
                            handed out chunk by chunk (video_utils.py:7-33; same indexing, including the 21st "truth" slot)
    ingest_item             uint8 [T,3,H,W,3] -> data [T,3,H/s,W/s,3] (nearest), target [T,1,H,W,3], high_frames [T,3,H,W,3]
                            as float32, ON THE DEVICE in one kernel (main.py:155-167, MakeData/Target/HFDatasetToTensor)
    run_item                `for x, y, high_frame in zip(data, target, high_frames): output, loss = model(x, y, high_frame,
                            estimated_image); estimated_image = output` (main.py:196-203) -> HR frames (+ losses)
    frames_to_u8            HR write-out, float32 -> uint8 NHWC on the device (the step after the path; the reference never
                            writes its frames)
    save_checkpoint / load_checkpoint   utils/tools.py:68-73 and main.py:108-122,233-237: {'arch','epoch','state_dict':
                            SRmodel.model.state_dict(),'optimizer'} -- files interchange with the reference's

`python -m video_super_resolution_amd.driver` is BASELINE.json's config C1 (3-frame 128x128 LR synthetic clip through the
main-like plumbing) on the GPU path (`tools/c1_check.py` runs the CPU checker beside it and reports the PSNR between the two:
the package itself never touches the checker).
Decoding compressed video (cv2.VideoCapture, video_utils.py:17-23) is out of scope: a clip enters as a uint8 RGB array
(`.npy`, raw rgb24, or synthetic).
"""
from __future__ import annotations

import os
import shutil
from glob import glob
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L


# ------------------------------------------------------------------------------------------------ dataset
def read_clip(path: str, shape: Optional[Tuple[int, int]] = None) -> np.ndarray:
    """A clip as uint8 RGB [T,H,W,3]: `.npy`, or headerless rgb24 (`shape` = (H, W) required)."""
    if path.endswith(".npy"):
        a = np.load(path)
    else:
        if shape is None:
            raise ValueError("raw rgb24 clips need shape=(H, W)")
        a = np.fromfile(path, dtype=np.uint8)
        a = a.reshape(-1, shape[0], shape[1], 3)
    if a.dtype != np.uint8 or a.ndim != 4 or a.shape[3] != 3:
        raise ValueError(f"{path}: expected uint8 [T,H,W,3], got {a.dtype} {a.shape}")
    return a


def sliding_windows(imgs: Sequence[np.ndarray], splitvideonum: int = 20) -> List[List[Sequence[np.ndarray]]]:
    """video_utils.py:24-27: 3-frame windows `imgs[i:i+3]`, cut into chunks of `length // splitvideonum` windows starting
    at every multiple of that step below `length` (length = FRAME count, so the last chunks are short or empty -- kept)."""
    length = len(imgs)
    data = [imgs[i:i + 3] for i in range(len(imgs) - 2)]
    step = int(length / splitvideonum)
    if step <= 0:
        raise ValueError(f"a clip needs at least {splitvideonum} frames (video_utils.py:26 steps by int(length / {splitvideonum}))")
    return [data[i:i + step] for i in range(0, length, step)]


class VideoDataset(torch.utils.data.Dataset):
    """utils/video_utils.py:7-33 without OpenCV: `videos` are paths (`.npy` / rgb24 + `raw_shape`), arrays [T,H,W,3] uint8,
    or a directory (globbed like the reference).  `__len__` = videos x 21, item `idx` with idx % 21 == 0 (re)reads video
    idx // 21, every item pops the next chunk: a list of 3-frame windows ([3,H,W,3] uint8 each)."""

    def __init__(self, videos, raw_shape: Optional[Tuple[int, int]] = None, splitvideonum: int = 20):
        if isinstance(videos, str):
            videos = sorted(glob(os.path.join(videos, "*")))
        elif isinstance(videos, np.ndarray):   # ONE clip [T,H,W,3], not a list of clips (list() would split it into frames)
            videos = [videos]
        self.video_paths = list(videos)
        self.raw_shape = raw_shape
        self.data: list = []
        self.splitvideonum = splitvideonum
        self.truthsplitvideonum = splitvideonum + 1

    def __len__(self):
        return len(self.video_paths) * self.truthsplitvideonum

    def read_video(self, v):
        imgs = read_clip(v, self.raw_shape) if isinstance(v, str) else np.asarray(v)
        self.data.extend(sliding_windows(list(imgs), self.splitvideonum))

    def __getitem__(self, idx):
        if idx % self.truthsplitvideonum == 0:
            self.read_video(self.video_paths[idx // self.truthsplitvideonum])
        data = self.data[0]
        self.data = self.data[1:]
        return data


# ------------------------------------------------------------------------------------------------ ingest / write-out
@L.on_device
def ingest_item(datas_u8: torch.Tensor, scale: int = 4, want_hr: bool = True):
    """uint8 [T,3,H,W,3] on the device -> (data [T,3,H//s,W//s,3], target [T,1,H,W,3] | None, high_frames [T,3,H,W,3] | None),
    float32 (main.py:155-167; `scale` is 4 there: `int(d.shape[1] / 4)`)."""
    if datas_u8.dtype != torch.uint8 or datas_u8.dim() != 5 or datas_u8.shape[1] != 3 or datas_u8.shape[4] != 3:
        raise ValueError(f"expected uint8 [T,3,H,W,3], got {datas_u8.dtype} {tuple(datas_u8.shape)}")
    T, _, H, W, _ = datas_u8.shape
    h, w = int(H / scale), int(W / scale)
    d = datas_u8.contiguous()
    lr = torch.empty((T, 3, h, w, 3), dtype=torch.float32, device=d.device)
    hr = torch.empty((T, 3, H, W, 3), dtype=torch.float32, device=d.device) if want_hr else None
    L.check(L.load().vsr_clip_ingest_u8(L.dptr(d, torch.uint8), L.dptr(lr), L.optr(hr), T * 3, H, W, h, w, L.stream()), "clip_ingest")
    if hr is None:
        return lr, None, None
    # target = datas[:, 1:2].float() (main.py:161-163): its own tensor -- VSR.forward overwrites high_frames[1] in place (:66)
    return lr, hr[:, 1:2].clone(), hr


@L.on_device
def frames_to_u8(frames: torch.Tensor) -> torch.Tensor:
    """float32 HR frames (any shape) -> uint8, round half to even, clamped to 0..255."""
    f = frames.detach().to(torch.float32).contiguous()
    out = torch.empty(f.shape, dtype=torch.uint8, device=f.device)
    import ctypes
    L.check(L.load().vsr_frame_to_u8(L.dptr(f), L.dptr(out, torch.uint8), ctypes.c_size_t(f.numel()), L.stream()), "frame_to_u8")
    return out


# ------------------------------------------------------------------------------------------------ the per-item loop
def run_item(model, data, target, high_frames, train: bool = False, estimated_image=None):
    """main.py:196-203 for one dataset item: windows in order, the output of one fed back as the next `estimated_image`.
    -> (outputs [T,H,W,3] float32, losses list, last estimate)."""
    outs, losses = [], []
    T = data.shape[0]
    with torch.no_grad():
        for t in range(T):
            x = data[t]
            y = target[t] if target is not None else None
            hf = high_frames[t] if high_frames is not None else None
            output, loss = model(x, y, hf, estimated_image, train=train)
            estimated_image = output
            outs.append(output[0])
            if loss is not None:
                losses.append(loss.data)
    return torch.stack(outs), losses, estimated_image


# ------------------------------------------------------------------------------------------------ checkpoints
def save_checkpoint(state, is_best, path, prefix, filename="checkpoint.pth.tar"):
    """utils/tools.py:68-73, same file naming."""
    prefix_save = os.path.join(path, prefix)
    name = prefix_save + "_" + filename
    torch.save(state, name)
    if is_best:
        shutil.copyfile(name, prefix_save + "_model_best.pth.tar")
    return name


def checkpoint_state(model, epoch: int, optimizer=None, arch: str = "VSR") -> dict:
    """The dict main.py:233-237 / :248-252 saves: only the SR net's state_dict is checkpointed."""
    return {"arch": arch, "epoch": epoch, "state_dict": model.model.state_dict(), "optimizer": optimizer}


def load_checkpoint(model, path: str, map_location=None, trusted: bool = False) -> dict:
    """main.py:108-122: `SRmodel.model.load_state_dict(checkpoint['state_dict'])` (strict).
    The file is first read with `weights_only=True` (tensors and plain containers only).  The reference's checkpoints hold
    the pickled Adam OBJECT under 'optimizer' (main.py:236), which that mode refuses: such a file is unpickled in full only
    when the caller says it is `trusted` (unpickling runs arbitrary code from the file)."""
    import pickle
    try:
        ckpt = torch.load(path, map_location=map_location, weights_only=True)
    except pickle.UnpicklingError as e:   # ONLY the weights-only refusal (an object beyond tensors / containers); a missing or
        #                                   corrupt file, a bad map_location etc. propagate as what they are
        if not trusted:
            raise RuntimeError(f"{path} holds pickled objects beyond tensors (the reference saves its optimizer object, "
                               f"main.py:236); pass trusted=True to unpickle it in full -- only for files you trust") from e
        ckpt = torch.load(path, map_location=map_location, weights_only=False)
    model.model.load_state_dict(ckpt["state_dict"])
    return ckpt


# ------------------------------------------------------------------------------------------------ config C1
def synthetic_video(n_frames: int, H: int, W: int, seed: int = 1234) -> np.ndarray:
    """uint8 RGB [T,H,W,3]: blurred-noise scene translated by (2k, k) px per frame (distribution S)."""
    from scipy.ndimage import gaussian_filter
    rs = np.random.RandomState(seed)
    pad = 4 * n_frames
    base = gaussian_filter(rs.uniform(0, 255, size=(H + pad, W + 2 * pad, 3)).astype(np.float32), sigma=(3, 3, 0))
    base = (base - base.min()) / (base.max() - base.min()) * 255.0
    return np.stack([np.floor(base[k:k + H, 2 * k:2 * k + W]) for k in range(n_frames)]).astype(np.uint8)


def run_c1(lr: int = 128, frames: int = 3, scale: int = 4, precision: str = "fp32"):
    """Config C1: a synthetic clip through ingest_item / run_item on the GPU path.
    -> (result line, model, datas uint8 [T,3,H,W,3] on the device, outputs [T,H,W,3])."""
    import time
    from . import VSR
    from .weights import fill_module_
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    S = scale
    model = fill_module_(VSR(upscale_factor=S).eval(), seed=0).to(dev)
    model.precision = model.model.precision = precision
    video = synthetic_video(frames, S * lr, S * lr)
    windows = [video[i:i + 3] for i in range(frames - 2)]                 # one dataset item (video_utils.py:25)
    datas = torch.from_numpy(np.stack(windows)).to(dev)                   # main.py:186 `torch.tensor(dataset[batch_idx])`
    data, target, high_frames = ingest_item(datas, S)
    run_item(model, data, target, high_frames)                            # warm-up (packing, allocator)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs, _, _ = run_item(model, data, target, high_frames)
    u8 = frames_to_u8(outs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    line = dict(config=f"C1: {frames}-frame {lr}x{lr} LR synthetic clip, x{S}, main-like driver, GPU path",
                precision=precision, windows=len(windows), frames_per_s=round(len(windows) / dt, 3), out_shape=list(u8.shape))
    return line, model, datas, outs


def main(argv=None):
    import argparse
    import json
    ap = argparse.ArgumentParser(description="config C1: a synthetic clip through the main.py-like plumbing on the GPU path")
    ap.add_argument("--lr", type=int, default=128, help="LR frame size (BASELINE.json C1: 128)")
    ap.add_argument("--frames", type=int, default=3, help="frames of the clip (3 = one window)")
    ap.add_argument("--scale", type=int, default=4, choices=[2, 3, 4], help="4 = the reference's geometry; 2 = C1's label")
    ap.add_argument("--precision", default="fp32", choices=["fp16", "fp32"])
    args = ap.parse_args(argv)
    line, _, _, _ = run_c1(args.lr, args.frames, args.scale, args.precision)
    print(json.dumps(line))


if __name__ == "__main__":
    main()
