"""Small helpers shared by the operator wrappers: raw pointers, current HIP stream, argument checks."""
import ctypes as C

import torch

from . import _lib


def ptr(t):
    """device pointer of a tensor (None -> NULL)"""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def stream():
    """hipStream_t of torch's current stream: kernels launch where the caller's torch ops do
    (the reference launches on the legacy default stream)."""
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_cuda(**tensors):
    """the reference's CHECK_CUDA / CHECK_CONTIGUOUS (gridencoder.cu:15-18) -> RuntimeError"""
    for name, t in tensors.items():
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(f"{name} must be a CUDA tensor")
        if not t.is_contiguous():
            raise RuntimeError(f"{name} must be a contiguous tensor")


def call(name, *args):
    _lib.call(name, *args)
