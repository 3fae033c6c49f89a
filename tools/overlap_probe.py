"""Can an HBM-bound BN-backward pass and an MFMA-bound weight-gradient kernel share the chip?  Times each alone and both
on two streams (level-0 shapes of cfg3).  Run on a GPU box: python tools/overlap_probe.py"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import uresnet_amd  # noqa: F401,E402
from uresnet_amd import _lib  # noqa: E402
from tests._ops import P, desc  # noqa: E402

lib = _lib.load()
N, S, C = 4, 192, 8
V = N * S ** 3
x = torch.randn(V, C, device="cuda"); dy = torch.randn(V, C, device="cuda"); z = torch.randn(V, C, device="cuda")
dz = torch.empty_like(z); y = torch.relu(z); dbeta = torch.zeros(C, device="cuda")
nb = lib.ursn_bn_scratch_bytes(V, C)
bn_scr = torch.empty(nb, dtype=torch.uint8, device="cuda")
d = desc(3, N, (S, S, S), C, C, 3, 1)
w = torch.zeros((3, 3, 3, C, C), device="cuda")
ws = lib.ursn_conv_wgrad_scratch_bytes(ctypes.byref(d))
w_scr = torch.empty(ws + 256, dtype=torch.uint8, device="cuda")
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
hs = lambda s: ctypes.c_void_p(s.cuda_stream)


def bn(s, reps):
    for _ in range(reps):
        _lib.check(lib.ursn_bn_backward(P(dy), P(y), P(z), P(dz), P(dbeta), V, C, 1e-3, 0, P(bn_scr), nb, hs(s)))


def wg(s, reps):
    for _ in range(reps):
        _lib.check(lib.ursn_conv_backward_weight(ctypes.byref(d), P(x), P(dy), P(w), P(w_scr), ws, hs(s)))


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


R = 5
bn(sa, 1); wg(sb, 1)
t_bn = timed(lambda: bn(sa, R)) / R
t_wg = timed(lambda: wg(sb, R)) / R
t_both = timed(lambda: (bn(sa, R), wg(sb, R))) / R
# interleaved issue order (as the net does: one BN pass, one weight gradient, ...)
def inter():
    for _ in range(R):
        bn(sa, 1); wg(sb, 1)
t_int = timed(inter) / R
# who stretches when both run?  per-stream durations of ONE concurrent pair
e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
torch.cuda.synchronize()
e[0].record(sa); bn(sa, 1); e[1].record(sa)
e[2].record(sb); wg(sb, 1); e[3].record(sb)
torch.cuda.synchronize()
print("concurrent pair: bn stream %.3f ms, wgrad stream %.3f ms" % (e[0].elapsed_time(e[1]), e[2].elapsed_time(e[3])))
print("bn_bwd alone %.3f ms, wgrad alone %.3f ms, sum %.3f; both streams %.3f (bulk issue) %.3f (interleaved issue)" % (
    t_bn, t_wg, t_bn + t_wg, t_both, t_int))
