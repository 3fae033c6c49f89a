import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from oracle import uresnet_np as O
from _ops import P, desc, stream
import uresnet_amd
from uresnet_amd import _lib
lib = _lib.load()
def bf(a):
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda().to(torch.bfloat16)
    return t.float().cpu().numpy().astype(np.float64), t
rng = np.random.default_rng(0)
ndim, N, S, ci, co, k, st = 3, 1, (2, 8, 32), 8, 8, 3, 1
x, xg = bf(rng.integers(-3, 4, (N,) + S + (ci,)).astype(np.float64))
for name, wfun in [("slot %d" % sl, (lambda sl: (lambda w: [w.__setitem__((sl // 9, (sl // 3) % 3, sl % 3, c, c), 1.0) for c in range(8)]))(sl)) for sl in range(27)] + [
                   ("random", None)]:
    w = np.zeros((3, 3, 3, ci, co))
    if wfun is None:
        w = rng.integers(-2, 3, w.shape).astype(np.float64)
    else:
        wfun(w)
    wg = torch.from_numpy(w.astype(np.float32)).cuda()
    y = O.conv_fwd(x, w, st)
    d = desc(ndim, N, S, ci, co, k, st); d.dtype = 1
    yg = torch.full(y.shape, float("nan"), dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.ursn_conv_forward(ctypes.byref(d), P(xg), P(wg), P(yg), stream()))
    torch.cuda.synchronize()
    got = yg.float().cpu().numpy()
    bad = np.argwhere(np.abs(got - y) > 1e-3)
    print(name, "max err", np.abs(got - y).max(), "n bad", len(bad), "of", y.size, "first bad", bad[:4].tolist())
    if len(bad):
        i = tuple(bad[0]); print("   got", got[i], "want", y[i], " got row", got[i[:-1]], " want row", y[i[:-1]])
