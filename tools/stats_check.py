"""Fused BatchNorm moments of the stride-2 / scatter kernels at realistic sizes against the exact moments of their own output."""
import ctypes, sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from _ops import P, desc, stream
import uresnet_amd
from uresnet_amd import _lib
lib = _lib.load()
torch.manual_seed(0)
for tag, ndim, N, S, ci, co, k, st, tr, algo in [("s2conv 2d", 2, 16, (512, 512), 16, 32, 3, 2, 0, 6), ("s2conv 3d", 3, 2, (96, 96, 96), 16, 32, 3, 2, 0, 6),
                                                   ("s2scatter 2d", 2, 16, (128, 128), 64, 32, 3, 2, 1, 7), ("s2scatter 3d", 3, 2, (48, 48, 48), 32, 16, 3, 2, 1, 7),
                                                   ("igemm 2d", 2, 16, (128, 128), 64, 64, 3, 1, 0, 4), ("tconv 2d", 2, 16, (512, 512), 16, 16, 3, 1, 0, 3),
                                                   ("pconv s2", 2, 16, (512, 512), 16, 32, 1, 2, 0, 5)]:
    x = torch.relu(torch.randn((N,) + S + (ci,), device="cuda")) * (torch.rand((N,) + S + (1,), device="cuda") > 0.97) * 20
    w = torch.randn((k,) * ndim + ((co, ci) if tr else (ci, co)), device="cuda") * 0.2
    d = desc(ndim, N, S, ci, co, k, st, transposed=tr, algo=algo)
    osp = tuple(2 * s for s in S) if tr else tuple((s + st - 1) // st for s in S)
    y = torch.empty((N,) + osp + (co,), device="cuda")
    mg, rg = torch.empty(co, device="cuda"), torch.empty(co, device="cuda")
    nb = 1 << 26
    scratch = torch.empty(nb, dtype=torch.uint8, device="cuda")
    _lib.check(lib.ursn_conv_forward_stats(ctypes.byref(d), P(x), P(w), P(y), P(mg), P(rg), 1e-3, P(scratch), nb, stream()))
    torch.cuda.synchronize()
    yd = y.double().reshape(-1, co)
    mu, var = yd.mean(0), yd.var(0, unbiased=False)
    e_mu = ((mg.double() - mu).abs() / var.sqrt()).max().item()
    e_rs = ((rg.double() - 1 / (var + 1e-3).sqrt()).abs() * (var + 1e-3).sqrt()).max().item()
    print("%-14s mean err %.2e std, rstd rel err %.2e   (|mean|/std max %.2f)" % (tag, e_mu, e_rs, (mu.abs() / var.sqrt()).max().item()))
