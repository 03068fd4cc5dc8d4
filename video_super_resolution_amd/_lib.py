"""ctypes binding of libvsr_hip.so (the C ABI declared in include/vsr_hip.h).

There is deliberately no CPU or eager-PyTorch fallback behind these entry points: if the
shared library is missing, or an operator is handed a non-CUDA tensor, the call raises.
"""
from __future__ import annotations

import ctypes
import os
import re
import subprocess

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
LIB_PATH = os.path.join(_PKG, "libvsr_hip.so")
XLIB_PATH = os.path.join(_PKG, "libvsr_hip_xcheck.so")
HEADER_PATH = os.path.join(_ROOT, "include", "vsr_hip.h")
XHEADER_PATH = os.path.join(_ROOT, "include", "vsr_hip_xcheck.h")
_lib = None
_xlib = None
# the cross-check library instead of the shipping one for every call (set by `xcheck()`; the environment switch serves the
# measurement tools, whose VSR_TUNING codes only that library understands)
_use_x = os.environ.get("VSR_USE_XCHECK", "0") == "1" or bool(os.environ.get("VSR_TUNING", "").strip())


class VsrHipError(RuntimeError):
    pass


def build(verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into the in-tree libvsr_hip.so and libvsr_hip_xcheck.so (hipcc cross-compiles without
    a GPU)."""
    cmd = ["make", "-j4", "-C", os.path.join(_PKG, "csrc")]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    return LIB_PATH


def declared_symbols(xcheck: bool = False) -> list:
    """Entry points include/vsr_hip.h declares (xcheck: the ones include/vsr_hip_xcheck.h adds)."""
    with open(XHEADER_PATH if xcheck else HEADER_PATH) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(vsr_[a-z0-9_]+)\s*\(", text)))


def _open(path: str) -> ctypes.CDLL:
    if not os.path.exists(path):
        raise VsrHipError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the device path)")
    lib = ctypes.CDLL(path)
    lib.vsr_last_error.restype = ctypes.c_char_p
    lib.vsr_last_route.restype = ctypes.c_char_p
    for fn in ("vsr_sr_query", "vsr_train_corr_dw_ws_floats", "vsr_train_prelu_bwd_ws_floats"):
        getattr(lib, fn).restype = ctypes.c_size_t
    if lib.vsr_abi_version() != 3:
        raise VsrHipError(f"{os.path.basename(path)}: ABI version mismatch")
    return lib


def load_xcheck() -> ctypes.CDLL:
    """libvsr_hip_xcheck.so (include/vsr_hip_xcheck.h): the shipping entry points + the superseded builds, switches and stamped
    diagnostics.  For tests and tools; the package calls it only from code paths that exist for the tests."""
    global _xlib
    if _xlib is None:
        lib = _open(XLIB_PATH)
        for t in os.environ.get("VSR_TUNING", "").split(","):   # measurement hook: kernel-selection switches (vsr_conv2d_tuning codes)
            if t.strip():
                lib.vsr_conv2d_tuning(int(t))
        _xlib = lib
    return _xlib


def load() -> ctypes.CDLL:
    """The library every product call goes through: libvsr_hip.so, unless a test / tool asked for the cross-check library."""
    global _lib
    if _use_x:
        return load_xcheck()
    if _lib is None:
        _lib = _open(LIB_PATH)
    return _lib


class xcheck:
    """`with _lib.xcheck():` -- every `load()` inside returns the cross-check library (so a switch set through it reaches the
    kernels the package's classes launch)."""

    def __enter__(self):
        global _use_x
        self._old, _use_x = _use_x, True
        return load_xcheck()

    def __exit__(self, *exc):
        global _use_x
        _use_x = self._old
        return False


# vsr_sr_query codes (include/vsr_hip.h)
Q_UTD_BLOB_BYTES, Q_UTD_STRIP_WIDTH, Q_UTD_S2_BLOB_BYTES, Q_UTD_S2_STRIP_WIDTH, Q_TAIL_S2_BLOB_BYTES = range(5)


def check(rc: int, what: str = "", lib=None) -> None:
    """`lib`: the library that made the call (each has its own error buffer); by default the text of every loaded library is
    shown, the one `load()` returns first (a call site that went through `load_xcheck()` directly is then still reported)."""
    if rc != 0:
        libs = [lib] if lib is not None else [l for l in (load(), _lib, _xlib) if l is not None]
        msgs = []
        for l in libs:
            m = l.vsr_last_error().decode("utf-8", "replace")
            if m and m not in msgs:
                msgs.append(m)
        raise VsrHipError(f"{what or 'vsr'} failed ({rc}): {' | '.join(msgs)}")


def dptr(t: torch.Tensor, dtype=torch.float32) -> ctypes.c_void_p:
    """Device pointer of a dense tensor; refuses anything the kernels' indexing does not assume."""
    if not t.is_cuda:
        raise VsrHipError("device path called with a CPU tensor (no CPU fallback exists)")
    if t.dtype != dtype:
        raise VsrHipError(f"expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise VsrHipError("tensor must be contiguous")
    return ctypes.c_void_p(t.data_ptr())


def optr(t, dtype=torch.float32):
    return ctypes.c_void_p(0) if t is None else dptr(t, dtype)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def raw_stream(index=None) -> int:
    """Handle of the current stream of device `index` (default: the current device) as an integer.  A forward issues ~900 launches and
    as many buffer look-ups keyed by the stream: `torch.cuda.current_stream()` builds a Stream object per call (~6 us of host time each),
    the raw accessor does not."""
    if _raw_stream is not None:
        return _raw_stream(torch._C._cuda_getDevice() if index is None else index)
    return torch.cuda.current_stream(index).cuda_stream


def stream() -> ctypes.c_void_p:
    """The current stream of the CURRENT device: entry points make the tensors' device current first (`on_device`)."""
    return ctypes.c_void_p(raw_stream())


def _first_cuda_tensor(objs):
    for o in objs:
        if isinstance(o, torch.Tensor):
            if o.is_cuda:
                return o
        elif isinstance(o, (list, tuple)):
            t = _first_cuda_tensor(o)
            if t is not None:
                return t
    return None


def on_device(fn):
    """Run `fn` with the device of its first CUDA tensor argument made current: kernels are launched on the current
    device's current stream and the >64 KB LDS attribute is kept per device, so a module living on cuda:1 must not launch
    while cuda:0 is current.  No-op (no context switch) when that device is current already."""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        t = _first_cuda_tensor(args)
        if t is None:
            t = _first_cuda_tensor(tuple(kwargs.values()))
        if t is None or t.device.index == torch.cuda.current_device():
            return fn(*args, **kwargs)
        with torch.cuda.device(t.device):
            return fn(*args, **kwargs)
    return wrapped


def cf(v: float) -> ctypes.c_float:
    return ctypes.c_float(float(v))


class _ChainStage(ctypes.Structure):
    _fields_ = [("inp", ctypes.c_void_p * 2), ("w", ctypes.c_void_p * 2), ("ldw", ctypes.c_int * 2), ("w_prev", ctypes.c_void_p),
                ("ldw_prev", ctypes.c_int), ("bias", ctypes.c_void_p), ("cmap", ctypes.c_void_p), ("slope", ctypes.c_float),
                ("out", ctypes.c_void_p)]


class Chain1x1(ctypes.Structure):
    """vsr_chain1x1_t of include/vsr_hip.h (all pointers as integers)."""
    _fields_ = [("nstages", ctypes.c_int), ("stage", _ChainStage * 3)]


_ROUTE_BATCH = os.environ.get("VSR_ROUTE_BATCH", "1") != "0"   # (A/B switch: 0 = every launcher follows the batch it is handed, as before round 5)


class route_batch:
    """`with _lib.route_batch(num, den):` -- the convolution launchers inside choose kernel / tile width / split-K as if their batch N
    were N * num / den (vsr_conv2d_route_batch): a trunk evaluated on a part of its usual batch runs the usual kernels.  num == den or
    den <= 0: no-op."""

    def __init__(self, num: int, den: int):
        self.num, self.den = int(num), int(den)
        self.on = self.den > 0 and self.num > 0 and self.num != self.den and _ROUTE_BATCH

    def __enter__(self):
        if self.on:
            check(load().vsr_conv2d_route_batch(self.num, self.den), "conv2d_route_batch")
        return self

    def __exit__(self, *exc):
        if self.on:
            load().vsr_conv2d_route_batch(0, 0)
        return False


class EventTimer:
    """HIP-event timing of selected kernel launches on the current stream (used by bench.py for the roofline
    leg).  Disabled (no events, no overhead) unless `enabled` is set."""

    def __init__(self):
        self.enabled = False
        self.only = None   # optional set of names: everything else stays untimed (bench.py times the dominant kernel only)
        self._pairs = {}

    def reset(self):
        self._pairs = {}

    def start(self, name: str):
        if not self.enabled or (self.only is not None and name not in self.only):
            return None
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()
        self._pairs.setdefault(name, []).append((a, b))
        return b

    @staticmethod
    def stop(tok):
        if tok is not None:
            tok.record()

    def summary(self):
        """{name: (launches, mean milliseconds)} -- call after a device synchronise."""
        return {k: (len(v), sum(a.elapsed_time(b) for a, b in v) / len(v)) for k, v in self._pairs.items()}


TIMER = EventTimer()


class RouteLog:
    """Which kernel every convolution launch was routed to (`vsr_last_route`), recorded while `enabled`: the launchers
    pick by layer size, so the full-size parity tests log what actually ran (label -> route -> count)."""

    def __init__(self):
        self.enabled = False
        self.calls = []

    def note(self, label: str):
        if self.enabled:
            self.calls.append((label, load().vsr_last_route().decode()))

    def note_as(self, label: str, route: str):
        """A launch whose route the caller names itself (the stock operator; a suffix on `vsr_last_route`)."""
        if self.enabled:
            self.calls.append((label, route))

    def last(self) -> str:
        return load().vsr_last_route().decode()

    def histogram(self):
        h = {}
        for _, r in self.calls:
            h[r] = h.get(r, 0) + 1
        return h


ROUTES = RouteLog()
