"""Times one bf16 conv layer (forward with statistics, data gradient, weight gradient) through the op-level C-ABI.
python tools/bf16_op_bench.py S cin cout [k stride transposed N]"""
import ctypes, sys, os, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from _ops import P, desc, stream
import uresnet_amd
from uresnet_amd import _lib
lib = _lib.load()
S, ci, co = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
k, st, tr, N = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((4, 3), (5, 1), (6, 0), (7, 4)))
d = desc(3, N, (S, S, S), ci, co, k, st, transposed=tr); d.dtype = 1
So = 2 * S if tr else (S + st - 1) // st
x = torch.randn((N, S, S, S, ci), device="cuda").to(torch.bfloat16)
w = torch.randn((k, k, k) + ((co, ci) if tr else (ci, co)), device="cuda") * 0.1
y = torch.empty((N, So, So, So, co), dtype=torch.bfloat16, device="cuda")
dy = torch.randn_like(y); dx = torch.empty_like(x); dw = torch.zeros_like(w)
mg, rg = torch.empty(co, device="cuda"), torch.empty(co, device="cuda")
nb = 1 << 26; scr = torch.empty(nb, dtype=torch.uint8, device="cuda")
wnb = lib.ursn_conv_wgrad_scratch_bytes(ctypes.byref(d)); wscr = torch.empty(wnb + 256, dtype=torch.uint8, device="cuda")
def t(fn, reps=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
f = (lambda: _lib.check(lib.ursn_conv_forward_stats(ctypes.byref(d), P(x), P(w), P(y), P(mg), P(rg), 1e-3, P(scr), nb, stream()))) if not tr else \
    (lambda: _lib.check(lib.ursn_conv_forward(ctypes.byref(d), P(x), P(w), P(y), stream())))
ms_f = t(f)
ms_d = t(lambda: _lib.check(lib.ursn_conv_backward_data(ctypes.byref(d), P(dy), P(w), P(dx), 0, stream())))
ms_w = t(lambda: _lib.check(lib.ursn_conv_backward_weight(ctypes.byref(d), P(x), P(dy), P(dw), P(wscr), wnb, stream())))
byt = (x.numel() + y.numel()) * 2
print("S=%d %d->%d k%d s%d t%d N=%d: fwd %.3f ms (%.0f GB/s) dgrad %.3f ms (%.0f GB/s) wgrad %.3f ms (%.0f GB/s)   env %s" % (
    S, ci, co, k, st, tr, N, ms_f, byt / ms_f / 1e6, ms_d, byt / ms_d / 1e6, ms_w, byt / ms_w / 1e6,
    {k: v for k, v in os.environ.items() if k.startswith("URSN_B")}))
