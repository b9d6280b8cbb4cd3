"""ctypes binding of libdtt_hip.so (include/dtt_hip.h).

There is no CPU fallback: if the library is missing, or a tensor is not on a
GPU, every op raises."""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# DTT_HIP_LIB: another build of the same library (A/B timing of kernel variants)
_LIB_PATH = os.environ.get("DTT_HIP_LIB") or os.path.join(_HERE, "libdtt_hip.so")
_lib = None


class DttError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise DttError(
                f"{_LIB_PATH} not found: build it with `make -C speech-separation_amd/csrc` "
                "(or __graft_entry__.build()). There is no fallback path.")
        _lib = C.CDLL(_LIB_PATH)
        _lib.dtt_last_error.restype = C.c_char_p
        _lib.dtt_version.restype = C.c_int
        _lib.dtt_gemm_ws_bytes.restype = C.c_int64
        _lib.dtt_conv3x3_wgrad_ws_bytes.restype = C.c_int64
        _lib.dtt_multi_pack_nchunks.restype = C.c_int64
    return _lib


def check(rc, what):
    if rc != 0:
        raise DttError(f"{what} failed ({rc}): {lib().dtt_last_error().decode()}")


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_RAW_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr():
    """The current stream of the current device as a hipStream_t.  Through torch's raw getters where they exist:
    torch.cuda.current_stream() builds a Stream object and resolves the device index in Python, 13 us a call --
    2.6 ms of host time per 1400-launch step in each of the forward and the backward thread."""
    if _RAW_STREAM is not None and _RAW_DEVICE is not None:
        return C.c_void_p(_RAW_STREAM(_RAW_DEVICE()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise DttError("dtt_amd ops need CUDA(HIP) tensors; there is no CPU path")
    return C.c_void_p(t.data_ptr())


def dtype_code(dt):
    if dt == torch.float32:
        return 0
    if dt == torch.bfloat16:
        return 1
    raise DttError(f"unsupported dtype {dt}")


class DttGemm(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("bias", C.c_void_p), ("R", C.c_void_p),
        ("a_ms", C.c_int64), ("a_ks", C.c_int64), ("a_bs", C.c_int64),
        ("b_ks", C.c_int64), ("b_ns", C.c_int64), ("b_bs", C.c_int64),
        ("c_ms", C.c_int64), ("c_ns", C.c_int64), ("c_bs", C.c_int64),
        ("r_ms", C.c_int64), ("r_ns", C.c_int64), ("r_bs", C.c_int64),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("batch", C.c_int32), ("splitk", C.c_int32),
        ("dtype", C.c_int32), ("out_f32", C.c_int32), ("atomic", C.c_int32), ("bias_mode", C.c_int32),
        ("act", C.c_int32), ("alpha", C.c_float),
        ("a_mode", C.c_int32), ("b_mode", C.c_int32),
        ("cB", C.c_int32), ("cFi", C.c_int32), ("cTi", C.c_int32), ("cCi", C.c_int32),
        ("cFo", C.c_int32), ("cTo", C.c_int32), ("cstride", C.c_int32), ("ctrans", C.c_int32),
        ("sh_shift", C.c_int32), ("sh_heads", C.c_int32), ("sh_L", C.c_int32), ("sh_edge", C.c_int32),
        ("cphase", C.c_int32), ("reserved_", C.c_int32),
        ("ws", C.c_void_p), ("ws_bytes", C.c_int64),
        ("cKW", C.c_int32), ("cPadF", C.c_int32), ("cPadT", C.c_int32), ("reserved2_", C.c_int32),
    ]


def gemm_raw(**kw):
    """Launch dtt_gemm from keyword fields (tensors for A/B/C/bias/R)."""
    keep = []
    for k in ("A", "B", "C", "bias", "R"):
        t = kw.get(k)
        if t is not None:
            if not t.is_cuda:
                raise DttError("dtt_amd ops need CUDA(HIP) tensors; there is no CPU path")
            keep.append(t)
            kw[k] = t.data_ptr()
        elif k in kw:
            del kw[k]
    kw.setdefault("batch", 1)
    kw.setdefault("splitk", 1)
    kw.setdefault("alpha", 1.0)
    g = DttGemm(**kw)                  # fields set by ctypes' initialiser, not by ~25 Python-level setattr calls
    if g.atomic and g.splitk > 1:
        # split-K partial tiles through a scratch buffer instead of fp32 atomics where the kernel can use one
        need = lib().dtt_gemm_ws_bytes(C.byref(g))
        if need > 0:
            import torch
            ws = torch.empty(need, dtype=torch.uint8, device=keep[0].device)
            keep.append(ws)
            g.ws, g.ws_bytes = ws.data_ptr(), need
    check(lib().dtt_gemm(C.byref(g), stream_ptr()), "dtt_gemm")
