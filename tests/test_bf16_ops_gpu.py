"""Op-level parity of the bf16 mixed-precision kernels (BASELINE.json configs[4]; bf16_conv.hip): bf16 tensors, fp32
weights / accumulation.  Oracle leg: the fp64 numpy oracle evaluated on inputs and weights ROUNDED TO bf16 (exactly what the
kernels read), so the only differences are the fp32 accumulation order (~1e-6) and the final rounding of the result to bf16
(half an ulp = 2^-9 relative per element).  Tolerances: bf16 outputs within 2^-8 of the tensor's max magnitude element-wise
(one bf16 ulp at the top of the range); fp32 weight gradients within 2e-5 of max.  PARITY UNPINNED (oracle/__init__.py)."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import uresnet_np as O
from _ops import P, desc, rel_err, stream
from uresnet_amd import _lib

pytestmark = pytest.mark.gpu
BF_TOL = 2.0 ** -8


def bf(a):
    """numpy fp64 -> values rounded to bf16 (returned as fp64) and the device bf16 tensor."""
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda().to(torch.bfloat16)
    return t.float().cpu().numpy().astype(np.float64), t


CASES = [
    # tag, ndim, N, S, cin, cout, k, stride, transposed
    ("k3s1_8_8", 3, 2, (8, 12, 40), 8, 8, 3, 1, 0),
    ("k3s1_8_8_odd", 3, 2, (9, 21, 37), 8, 8, 3, 1, 0),         # input-stationary kernel (bf16_conv3.hip): ragged tiles
    ("k3s1_8_8_zseg", 3, 1, (40, 18, 34), 8, 8, 3, 1, 0),       # ... several z segments per column of tiles
    ("k3s1_16_8_odd", 3, 2, (11, 19, 35), 16, 8, 3, 1, 0),      # ... 16 contraction channels (two pieces per staged voxel)
    ("k3s1_16_16_zseg", 3, 1, (36, 9, 40), 16, 16, 3, 1, 0),    # ... 16 produced channels (two row tiles, 32 x 8 voxel tiles)
    ("k3s1_8_16", 3, 1, (6, 10, 33), 8, 16, 3, 1, 0),
    ("k3s1_16_16", 3, 2, (4, 8, 32), 16, 16, 3, 1, 0),
    ("k3s1_32_32", 3, 1, (6, 6, 18), 32, 32, 3, 1, 0),
    ("k3s1_64_32", 3, 1, (4, 6, 12), 64, 32, 3, 1, 0),
    ("k3s1_64_128", 3, 1, (4, 4, 8), 64, 128, 3, 1, 0),
    ("k3s1_24_40", 3, 1, (4, 6, 20), 24, 40, 3, 1, 0),          # channel counts that are only multiples of 8
    ("k3s2_8_16", 3, 2, (8, 12, 36), 8, 16, 3, 2, 0),
    ("k3s2_odd", 3, 1, (7, 9, 21), 16, 32, 3, 2, 0),            # odd sizes: TF SAME pad-before = 1
    ("k1s1_16_8", 3, 2, (7, 9, 37), 16, 8, 1, 1, 0),            # 1x1 between 8 / 16 channels: operands straight from global memory
    ("k1s1_32_16", 3, 2, (6, 6, 20), 32, 16, 1, 1, 0),
    ("k1s2_8_16", 3, 2, (8, 12, 36), 8, 16, 1, 2, 0),
    ("deconv_16_8", 3, 2, (4, 6, 18), 16, 8, 3, 2, 1),
    ("deconv_64_32", 3, 1, (3, 4, 6), 64, 32, 3, 2, 1),
    ("2d_k3s1_16_16", 2, 2, (24, 70), 16, 16, 3, 1, 0),
    ("2d_k3s2_16_32", 2, 2, (24, 70), 16, 32, 3, 2, 0),
    ("2d_deconv_32_16", 2, 1, (12, 20), 32, 16, 3, 2, 1),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_bf16_conv_forward_data_and_weight_gradients(case):
    tag, ndim, N, S, ci, co, k, st, tr = case
    lib = _lib.load()
    rng = np.random.default_rng(len(tag) * 7 + ci)
    x, xg = bf(rng.standard_normal((N,) + S + (ci,)))
    wshape = (k,) * ndim + ((co, ci) if tr else (ci, co))
    w = rng.standard_normal(wshape) * 0.2
    w = torch.from_numpy(w.astype(np.float32)).to(torch.bfloat16).float().numpy().astype(np.float64)   # bf16-representable
    wg = torch.from_numpy(w.astype(np.float32)).cuda()
    y = O.deconv_fwd(x, w) if tr else O.conv_fwd(x, w, st)
    dy, dyg = bf(rng.standard_normal(y.shape))
    dx, dw = O.deconv_bwd(x, w, dy) if tr else O.conv_bwd(x, w, st, dy)
    d = desc(ndim, N, S, ci, co, k, st, transposed=tr)
    d.dtype = 1
    yg = torch.full(y.shape, float("nan"), dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.ursn_conv_forward(ctypes.byref(d), P(xg), P(wg), P(yg), stream()))
    torch.cuda.synchronize()
    e = np.abs(yg.float().cpu().numpy() - y).max() / np.abs(y).max()
    assert e <= BF_TOL, ("forward", e)
    dxg = torch.full(x.shape, float("nan"), dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.ursn_conv_backward_data(ctypes.byref(d), P(dyg), P(wg), P(dxg), 0, stream()))
    torch.cuda.synchronize()
    e = np.abs(dxg.float().cpu().numpy() - dx).max() / np.abs(dx).max()
    assert e <= BF_TOL, ("dgrad", e)
    base, baseg = bf(rng.standard_normal(x.shape))
    _lib.check(lib.ursn_conv_backward_data(ctypes.byref(d), P(dyg), P(wg), P(baseg), 1, stream()))
    torch.cuda.synchronize()
    e = np.abs(baseg.float().cpu().numpy() - (dx + base)).max() / np.abs(dx + base).max()
    assert e <= BF_TOL, ("dgrad accumulate", e)
    nb = lib.ursn_conv_wgrad_scratch_bytes(ctypes.byref(d))
    scratch = torch.empty(nb + 256, dtype=torch.uint8, device="cuda")
    dwg = torch.zeros(wshape, dtype=torch.float32, device="cuda")
    for rep in (1, 2):   # dw accumulates
        _lib.check(lib.ursn_conv_backward_weight(ctypes.byref(d), P(xg), P(dyg), P(dwg), P(scratch), nb, stream()))
        torch.cuda.synchronize()
        assert rel_err(dwg.cpu().numpy(), rep * dw) < 2e-5, ("wgrad", rep)


@pytest.mark.parametrize("case", [c for c in CASES if not c[8]][:16], ids=[c[0] for c in CASES if not c[8]][:16])
def test_bf16_conv_forward_fused_statistics(case):
    tag, ndim, N, S, ci, co, k, st, tr = case
    lib = _lib.load()
    rng = np.random.default_rng(len(tag) + co)
    x, xg = bf(rng.standard_normal((N,) + S + (ci,)) + 3.0)
    w = rng.standard_normal((k,) * ndim + (ci, co)) * 0.2
    wg = torch.from_numpy(w.astype(np.float32)).cuda()
    d = desc(ndim, N, S, ci, co, k, st)
    d.dtype = 1
    yshape = O.conv_fwd(x, w, st).shape
    yg = torch.empty(yshape, dtype=torch.bfloat16, device="cuda")
    mg, rg = torch.empty(co, device="cuda"), torch.empty(co, device="cuda")
    nb = 1 << 24
    scratch = torch.empty(nb, dtype=torch.uint8, device="cuda")
    _lib.check(lib.ursn_conv_forward_stats(ctypes.byref(d), P(xg), P(wg), P(yg), P(mg), P(rg), 1e-3, P(scratch), nb, stream()))
    torch.cuda.synchronize()
    ys = yg.float().cpu().numpy().astype(np.float64)      # the statistics are those of the STORED bf16 tensor
    ax = tuple(range(ys.ndim - 1))
    mu, var = ys.mean(axis=ax), ys.var(axis=ax)
    assert np.abs(mg.cpu().numpy() - mu).max() < 1e-5 * np.sqrt(var.max()) + 1e-6 * np.abs(mu).max()
    assert rel_err(rg.cpu().numpy(), 1 / np.sqrt(var + 1e-3)) < 1e-5
