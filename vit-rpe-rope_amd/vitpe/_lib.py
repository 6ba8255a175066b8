"""ctypes binding of libvitpe.so (the C ABI declared in include/vitpe.h).

The prototypes are parsed from the header itself, so the header is the single source of
truth for the boundary.  There is no fallback: if the library is missing, importing the
kernels fails loudly (the product path never routes through a CPU implementation).
"""
from __future__ import annotations

import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(_HERE)                       # vit-rpe-rope_amd/
REPO_ROOT = os.path.dirname(PKG_ROOT)
LIB_PATH = os.path.join(PKG_ROOT, "lib", "libvitpe.so")
HEADER_PATH = os.path.join(REPO_ROOT, "include", "vitpe.h")
DEBUG_HEADER_PATH = os.path.join(REPO_ROOT, "include", "vitpe_debug.h")   # self-tests / census: not the product ABI

F32, BF16 = 0, 1
PE_CODES = {"none": 0, "absolute": 1, "relative": 2, "polynomial": 3, "rope-axial": 4, "rope-mixed": 5}
EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESID, EPI_PATCH, EPI_GELU_BWD = range(5)


class VitpeError(RuntimeError):
    pass


_CTYPE = {"int": ctypes.c_int, "float": ctypes.c_float, "long long": ctypes.c_longlong}


def parse_header(path: str = HEADER_PATH):
    """-> {name: [ctypes argtypes]} for every `int vitpe_*(...)` declaration."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\bint\s+(vitpe_\w+)\s*\(([^)]*)\)\s*;", src):
        name, args = m.group(1), m.group(2).strip()
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a or a.startswith("vitpe_stream_t"):
                    argtypes.append(ctypes.c_void_p)
                else:
                    base = re.sub(r"\bconst\b", "", a).strip()
                    base = re.sub(r"\s+\w+$", "", base).strip()  # drop the parameter name
                    argtypes.append(_CTYPE[base])
        protos[name] = argtypes
    return protos


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VitpeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(python -c 'import __graft_entry__ as g; g.build()').  There is no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, argtypes in parse_header().items():
            fn = getattr(handle, name)  # AttributeError if the header declares a missing symbol
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
        _lib = handle
    return _lib


def debug_lib():
    """The same library with the developer entry points of include/vitpe_debug.h bound as well (tests / tools only)."""
    handle = lib()
    if not getattr(handle, "_vitpe_debug_bound", False):
        for name, argtypes in parse_header(DEBUG_HEADER_PATH).items():
            fn = getattr(handle, name)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
        handle._vitpe_debug_bound = True
    return handle


def check(err: int, what: str):
    if err != 0:
        name = {1: "hipErrorInvalidValue", 801: "hipErrorNotSupported", 98: "hipErrorInvalidDeviceFunction",
                719: "hipErrorLaunchFailure", 2: "hipErrorOutOfMemory"}.get(err, "hipError")
        raise VitpeError(f"{what} failed: {name} ({err})")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    raise VitpeError(f"unsupported compute dtype {dt}; use torch.float32 or torch.bfloat16")


def ptr(t):
    return None if t is None else t.data_ptr()


def require_device(*tensors):
    """The HIP path is the only path: refuse CPU tensors instead of silently falling back."""
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise VitpeError("vitpe: HIP device tensor required (got a CPU tensor; there is no CPU fallback)")
        if not t.is_contiguous():
            raise VitpeError("vitpe: contiguous tensor required")
