"""ctypes front-end of oracle/native_ops.c (checker only, see oracle/__init__.py)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_native.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "native_ops.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "_build/liboracle_native.so"])
    return _SO


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        fp = ctypes.POINTER(ctypes.c_float)
        ip = ctypes.POINTER(ctypes.c_int)
        _lib.oracle_resample2d.argtypes = [fp, fp, fp] + [ctypes.c_int] * 8
        _lib.oracle_resample2d.restype = None
        _lib.oracle_channelnorm.argtypes = [fp, fp] + [ctypes.c_int] * 4
        _lib.oracle_channelnorm.restype = None
        _lib.oracle_correlation_out_shape.argtypes = [ctypes.c_int] * 7 + [ip, ip, ip]
        _lib.oracle_correlation_out_shape.restype = ctypes.c_int
        _lib.oracle_correlation.argtypes = [fp, fp, fp] + [ctypes.c_int] * 9
        _lib.oracle_correlation.restype = ctypes.c_int
    return _lib


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def resample2d(img: np.ndarray, flow: np.ndarray, kernel_size: int = 1, bilinear: bool = True) -> np.ndarray:
    img, pi = _f32(img)
    flow, pf = _f32(flow)
    B, C, Hi, Wi = img.shape
    Bf, two, H, W = flow.shape
    assert two == 2 and Bf == B
    out = np.zeros((B, C, H, W), np.float32)
    lib().oracle_resample2d(pi, pf, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                            B, C, H, W, Hi, Wi, int(kernel_size), int(bool(bilinear)))
    return out


def channelnorm(x: np.ndarray) -> np.ndarray:
    x, px = _f32(x)
    B, C, H, W = x.shape
    out = np.zeros((B, 1, H, W), np.float32)
    lib().oracle_channelnorm(px, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), B, C, H, W)
    return out


def correlation_out_shape(H, W, pad_size, kernel_size, max_displacement, stride1, stride2):
    oc, oh, ow = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    rc = lib().oracle_correlation_out_shape(H, W, pad_size, kernel_size, max_displacement, stride1, stride2,
                                            ctypes.byref(oc), ctypes.byref(oh), ctypes.byref(ow))
    if rc:
        raise ValueError("correlation: empty output")
    return oc.value, oh.value, ow.value


def correlation(f1: np.ndarray, f2: np.ndarray, pad_size=20, kernel_size=1, max_displacement=20,
                stride1=1, stride2=2) -> np.ndarray:
    f1, p1 = _f32(f1)
    f2, p2 = _f32(f2)
    B, C, H, W = f1.shape
    assert f2.shape == f1.shape
    oc, oh, ow = correlation_out_shape(H, W, pad_size, kernel_size, max_displacement, stride1, stride2)
    out = np.zeros((B, oc, oh, ow), np.float32)
    rc = lib().oracle_correlation(p1, p2, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), B, C, H, W,
                                  pad_size, kernel_size, max_displacement, stride1, stride2)
    if rc:
        raise ValueError("correlation failed")
    return out
