"""ctypes binding of libvipformer_hip.so (the C ABI in include/vipformer_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails,
this raises.  (The CPU oracle under oracle/ is test infrastructure and is never
imported from here.)
"""
from __future__ import annotations

import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvipformer_hip.so")
_lib = None

VP = ctypes.c_void_p
I = ctypes.c_int
F = ctypes.c_float
U64 = ctypes.c_uint64


class VpfError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VpfError(
                f"{LIB_PATH} not found: build it with `python -m vipformer_amd.build` "
                "(vipformer_amd has no CPU/eager fallback)")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.vpf_strerror.restype = ctypes.c_char_p
        _lib.vpf_strerror.argtypes = [I]
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise VpfError(f"{what} failed: {lib().vpf_strerror(rc).decode()} (rc={rc})")


def stream_ptr() -> VP:
    return VP(torch.cuda.current_stream().cuda_stream)


def ptr(t) -> VP:
    if t is None:
        return VP(0)
    return VP(t.data_ptr())


def need_cuda(*ts) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise VpfError("vipformer_amd ops run on an MI355X only (got a CPU tensor); there is no CPU fallback")
