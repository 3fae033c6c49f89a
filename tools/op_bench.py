"""Runs one conv layer shape through the op-level C-ABI a few times (for rocprofv3 --pmc passes and quick A/B timing).
usage: python tools/op_bench.py NDIM N S CIN COUT K STRIDE TRANSPOSED ALGO [reps]
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import uresnet_amd  # noqa: F401,E402
from uresnet_amd import _lib  # noqa: E402
from tests._ops import P, desc, stream  # noqa: E402


def main():
    ndim, N, S, ci, co, k, st, tr, algo = [int(v) for v in sys.argv[1:10]]
    reps = int(sys.argv[10]) if len(sys.argv) > 10 else 3
    sp = (S,) * ndim
    lib = _lib.load()
    d = desc(ndim, N, sp, ci, co, k, st, transposed=tr, algo=algo)
    osp = tuple(2 * s for s in sp) if tr else tuple((s + st - 1) // st for s in sp)
    x = torch.randn((N,) + sp + (ci,), device="cuda")
    y = torch.empty((N,) + osp + (co,), device="cuda")
    dy = torch.randn_like(y)
    dx = torch.empty_like(x)
    wshape = (k,) * ndim + ((co, ci) if tr else (ci, co))
    w = torch.randn(wshape, device="cuda") * 0.1
    dw = torch.zeros_like(w)
    nb = lib.ursn_conv_wgrad_scratch_bytes(ctypes.byref(d))
    scratch = torch.empty(nb + 256, dtype=torch.uint8, device="cuda")
    macs = N * (x.numel() // N // ci if tr else y.numel() // N // co) * k ** ndim * ci * co
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for name, fn in (("fwd", lambda: lib.ursn_conv_forward(ctypes.byref(d), P(x), P(w), P(y), stream())),
                     ("dgrad", lambda: lib.ursn_conv_backward_data(ctypes.byref(d), P(dy), P(w), P(dx), 0, stream())),
                     ("wgrad", lambda: lib.ursn_conv_backward_weight(ctypes.byref(d), P(x), P(dy), P(dw), P(scratch), nb, stream()))):
        try:
            _lib.check(fn())
        except Exception as e:  # pass not supported by the forced algo
            print(name, "unsupported:", str(e)[:80])
            continue
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(reps):
            _lib.check(fn())
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / reps
        print("%-6s %.3f ms  %.1f TFLOP/s" % (name, ms, 2 * macs / ms / 1e9))


if __name__ == "__main__":
    main()
