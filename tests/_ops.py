"""Thin test helpers calling the op-level C-ABI on torch (ROCm) tensors."""
import ctypes

import numpy as np
import torch

import uresnet_amd  # noqa: F401
from uresnet_amd import _lib


def P(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def desc(ndim, n, in_sp, cin, cout, k, stride, transposed=0, in_cs=0, out_cs=0, algo=0):
    d = _lib.ursn_conv_desc()
    d.ndim, d.n, d.cin, d.cout, d.k, d.stride, d.transposed = ndim, n, cin, cout, k, stride, transposed
    for i in range(3):
        d.in_sp[i] = in_sp[i] if i < ndim else 1
    d.in_cstride, d.out_cstride, d.algo = in_cs, out_cs, algo
    return d


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def conv_forward(d, x, w, out_shape):
    lib = _lib.load()
    y = torch.full(out_shape, float("nan"), dtype=torch.float32, device="cuda")
    _lib.check(lib.ursn_conv_forward(ctypes.byref(d), P(x), P(w), P(y), stream()))
    torch.cuda.synchronize()
    return y


def conv_backward_data(d, dy, w, dx_shape, accumulate=0, dx_init=None):
    lib = _lib.load()
    dx = dx_init.clone() if dx_init is not None else torch.full(dx_shape, float("nan"), dtype=torch.float32, device="cuda")
    _lib.check(lib.ursn_conv_backward_data(ctypes.byref(d), P(dy), P(w), P(dx), accumulate, stream()))
    torch.cuda.synchronize()
    return dx


def conv_backward_weight(d, x, dy, w_shape, dw_init=None):
    lib = _lib.load()
    dw = dw_init.clone() if dw_init is not None else torch.zeros(w_shape, dtype=torch.float32, device="cuda")
    nb = lib.ursn_conv_wgrad_scratch_bytes(ctypes.byref(d))
    scratch = torch.empty(nb + 256, dtype=torch.uint8, device="cuda")
    _lib.check(lib.ursn_conv_backward_weight(ctypes.byref(d), P(x), P(dy), P(dw), P(scratch), nb, stream()))
    torch.cuda.synchronize()
    return dw


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
