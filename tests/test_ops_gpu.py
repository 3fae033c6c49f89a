"""Op-level parity: every HIP kernel family against the numpy fp64 oracle (PARITY UNPINNED: the
oracle is a restatement, see oracle/__init__.py).  Tolerances: fp32 accumulation over at most a
few thousand terms -> 2e-5 relative to the tensor's max magnitude."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import uresnet_np as O
from _ops import (P, conv_backward_data, conv_backward_weight, conv_forward, desc, dev, rel_err, stream)
from uresnet_amd import _lib

pytestmark = pytest.mark.gpu
TOL = 2e-5

CONV_CASES = [
    # ndim, N, S, cin, cout, k, stride
    (3, 2, (8, 12, 16), 8, 8, 3, 1),
    (3, 1, (8, 8, 8), 1, 8, 3, 1),
    (3, 2, (8, 8, 16), 8, 3, 3, 1),
    (3, 2, (8, 12, 16), 8, 16, 3, 2),
    (3, 2, (8, 12, 16), 8, 16, 1, 2),
    (3, 2, (6, 6, 6), 32, 16, 1, 1),
    (3, 1, (6, 6, 6), 64, 64, 3, 1),
    (3, 2, (2, 2, 2), 128, 256, 3, 2),
    (3, 1, (1, 1, 1), 16, 16, 3, 1),
    (2, 2, (16, 24), 16, 16, 3, 1),
    (2, 3, (16, 16), 16, 32, 3, 2),
    (2, 2, (16, 16), 1, 16, 3, 1),
    (2, 2, (8, 8), 16, 5, 3, 1),
    (2, 2, (4, 4), 256, 512, 3, 2),
    (2, 1, (8, 8), 32, 16, 1, 1),
    (3, 1, (4, 10, 20), 12, 20, 3, 1),   # channel counts that are not powers of two
]


def _rand(rng, shape):
    return rng.standard_normal(shape)


@pytest.mark.parametrize("algo", [1, 2])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_bwd(case, algo):
    ndim, N, S, ci, co, k, s = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = _rand(rng, (N,) + S + (ci,))
    w = _rand(rng, (k,) * ndim + (ci, co)) * 0.2
    y = O.conv_fwd(x, w, s)
    dy = _rand(rng, y.shape)
    dx, dw = O.conv_bwd(x, w, s, dy)
    d = desc(ndim, N, S, ci, co, k, s, algo=algo)
    xg, wg, dyg = dev(x), dev(w), dev(dy)
    yg = conv_forward(d, xg, wg, y.shape)
    assert rel_err(yg.cpu().numpy(), y) < TOL
    dxg = conv_backward_data(d, dyg, wg, x.shape)
    assert rel_err(dxg.cpu().numpy(), dx) < TOL
    # accumulate mode adds onto existing contents
    base = torch.ones(x.shape, dtype=torch.float32, device="cuda")
    dxa = conv_backward_data(d, dyg, wg, x.shape, accumulate=1, dx_init=base)
    assert rel_err(dxa.cpu().numpy(), dx + 1.0) < TOL
    dwg = conv_backward_weight(d, xg, dyg, w.shape)
    assert rel_err(dwg.cpu().numpy(), dw) < 5e-5
    # dw accumulates (assign_add semantics)
    dwa = conv_backward_weight(d, xg, dyg, w.shape, dw_init=dwg)
    assert rel_err(dwa.cpu().numpy(), 2 * dw) < 5e-5


DECONV_CASES = [
    (3, 2, (4, 6, 8), 16, 8),
    (3, 1, (1, 1, 1), 32, 16),
    (3, 2, (2, 2, 2), 256, 128),
    (2, 2, (8, 8), 32, 16),
    (2, 1, (2, 2), 512, 256),
]


@pytest.mark.parametrize("algo", [1, 2])
@pytest.mark.parametrize("case", DECONV_CASES)
def test_deconv_fwd_bwd(case, algo):
    ndim, N, S, ci, co = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = _rand(rng, (N,) + S + (ci,))
    w = _rand(rng, (3,) * ndim + (co, ci)) * 0.2
    y = O.deconv_fwd(x, w)
    dy = _rand(rng, y.shape)
    dx, dw = O.deconv_bwd(x, w, dy)
    d = desc(ndim, N, S, ci, co, 3, 2, transposed=1, algo=algo)
    xg, wg, dyg = dev(x), dev(w), dev(dy)
    assert rel_err(conv_forward(d, xg, wg, y.shape).cpu().numpy(), y) < TOL
    assert rel_err(conv_backward_data(d, dyg, wg, x.shape).cpu().numpy(), dx) < TOL
    assert rel_err(conv_backward_weight(d, xg, dyg, w.shape).cpu().numpy(), dw) < 5e-5


def test_conv_channel_strides():
    """Concat-free buffers: input read from / output written into channel slices of wider tensors."""
    rng = np.random.default_rng(5)
    N, S, ci, co = 2, (4, 6, 8), 8, 8
    xfull = _rand(rng, (N,) + S + (16,))
    w = _rand(rng, (3, 3, 3, ci, co)) * 0.2
    x = xfull[..., 8:]
    y = O.conv_fwd(x, w, 1)
    d = desc(3, N, S, ci, co, 3, 1, in_cs=16, out_cs=24, algo=2)
    xg = dev(xfull)
    yfull = torch.zeros((N,) + S + (24,), dtype=torch.float32, device="cuda")
    lib = _lib.load()
    xoff = ctypes.c_void_p(xg.data_ptr() + 8 * 4)
    yoff = ctypes.c_void_p(yfull.data_ptr() + 4 * 4)
    wg = dev(w)
    _lib.check(lib.ursn_conv_forward(ctypes.byref(d), xoff, P(wg), yoff, stream()))
    torch.cuda.synchronize()
    got = yfull.cpu().numpy()
    assert rel_err(got[..., 4:12], y) < TOL
    assert np.all(got[..., :4] == 0) and np.all(got[..., 12:] == 0)


@pytest.mark.parametrize("C,relu,res", [(8, 1, False), (16, 1, True), (3, 0, False), (64, 0, True), (12, 1, False)])
def test_bn_forward_backward(C, relu, res):
    rng = np.random.default_rng(C)
    V = 4 * 6 * 10 * 3
    z = _rand(rng, (V, C)) * 2.0 + 0.5
    beta = _rand(rng, (C,)) * 0.3
    r = _rand(rng, (V, C)) if res else None
    y, cache = O.bn_fwd(z, beta)
    if res:
        y = y + r
    if relu:
        y = np.maximum(y, 0)
    lib = _lib.load()
    nb = lib.ursn_bn_scratch_bytes(V, C)
    scratch = torch.empty(nb, dtype=torch.uint8, device="cuda")
    zg, bg = dev(z), dev(beta)
    rg = dev(r) if res else None
    yg = torch.empty((V, C), dtype=torch.float32, device="cuda")
    st = torch.empty(2 * C, dtype=torch.float32, device="cuda")
    _lib.check(lib.ursn_bn_forward(P(zg), P(bg), P(rg), P(yg), V, C, 1e-3, relu, P(st), P(scratch), nb, stream()))
    torch.cuda.synchronize()
    assert rel_err(yg.cpu().numpy(), y) < 1e-5
    assert rel_err(st.cpu().numpy()[C:], cache[1]) < 1e-5
    dy = _rand(rng, (V, C))
    g = dy * (y > 0) if relu else dy
    dz, dbeta = O.bn_bwd(cache, g)
    dzg = torch.empty((V, C), dtype=torch.float32, device="cuda")
    dbg = torch.zeros(C, dtype=torch.float32, device="cuda")
    dyg = dev(dy)
    _lib.check(lib.ursn_bn_backward(P(dyg), P(yg), P(zg), P(dzg), P(dbg), V, C, 1e-3, relu, P(scratch), nb,
                                    stream()))
    torch.cuda.synchronize()
    assert rel_err(dzg.cpu().numpy(), dz) < 2e-5
    assert rel_err(dbg.cpu().numpy(), dbeta) < 2e-5


@pytest.mark.parametrize("ncls,use_w", [(3, True), (5, False), (3, False)])
def test_softmax_ce_head(ncls, use_w):
    rng = np.random.default_rng(ncls)
    n, pix = 3, 500
    logits = _rand(rng, (n, pix, ncls)) * 3
    logits[0, :7] = 1.25  # exact ties -> argmax must return the lowest index
    data = rng.uniform(0, 1, (n, pix)) * (rng.uniform(0, 1, (n, pix)) > 0.6)
    label = rng.integers(0, ncls, (n, pix)).astype(np.float64)
    w = rng.uniform(0.1, 2, (n, pix)) if use_w else None
    m = O.loss_and_metrics(logits, data, label, w)
    lib = _lib.load()
    scratch = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    sm = torch.empty((n, pix, ncls), dtype=torch.float32, device="cuda")
    dl = torch.empty((n, pix, ncls), dtype=torch.float32, device="cuda")
    out = (ctypes.c_float * 3)()
    zg, dg, lg = dev(logits), dev(data), dev(label)  # keep references: P() only holds the raw pointer
    wg = dev(w) if use_w else None
    _lib.check(lib.ursn_softmax_ce(P(zg), P(dg), P(lg), P(wg), n, pix, ncls, P(sm), P(dl), out, P(scratch), 1 << 20,
                                   stream()))
    assert abs(out[0] - m["loss"]) / abs(m["loss"]) < 1e-5
    assert abs(out[1] - m["acc_all"]) < 1e-6
    assert abs(out[2] - m["acc_nonzero"]) < 1e-6
    assert rel_err(sm.cpu().numpy(), m["softmax"]) < 1e-5
    assert rel_err(dl.cpu().numpy(), m["dlogits"]) < 1e-5


def test_head_no_nonzero_pixels_gives_nan():
    lib = _lib.load()
    n, pix, ncls = 1, 64, 3
    scratch = torch.empty(1 << 16, dtype=torch.uint8, device="cuda")
    out = (ctypes.c_float * 3)()
    z = torch.zeros((n, pix, ncls), device="cuda")
    d0, l0 = torch.zeros(n, pix, device="cuda"), torch.zeros(n, pix, device="cuda")
    _lib.check(lib.ursn_softmax_ce(P(z), P(d0), P(l0), None, n, pix, ncls, None, None, out, P(scratch), 1 << 16,
                                   stream()))
    assert np.isnan(out[2]) and out[1] == 1.0 and abs(out[0] - pix * np.log(3)) < 1e-3


def test_adam_tf_form():
    rng = np.random.default_rng(0)
    n = 10007
    p = {"a": rng.standard_normal(n)}
    opt = O.Adam(p, lr=1e-3)
    pg, mg, vg = dev(p["a"]), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    lib = _lib.load()
    for t in range(1, 4):
        g = rng.standard_normal(n) * 10.0 ** rng.integers(-6, 2, n)
        opt.apply(p, {"a": g})
        gg = dev(g)
        _lib.check(lib.ursn_adam(P(pg), P(gg), P(mg), P(vg), n, 1e-3, 0.9, 0.999, 1e-8, t, stream()))
        torch.cuda.synchronize()
    torch.cuda.synchronize()
    assert rel_err(pg.cpu().numpy(), p["a"]) < 1e-6


def test_mfma_probe_layouts():
    """Pins the lane layouts the kernels assume (v_mfma_f32_16x16x4_f32 documented in the CDNA4
    guide; v_mfma_f32_4x4x1_16b_f32 incl. cbsz/abid broadcast decoded here)."""
    lib = _lib.load()
    out = torch.zeros(256, dtype=torch.float32, device="cuda")
    res = {}
    for which in range(1, 7):
        _lib.check(lib.ursn_mfma_probe(which, P(out), stream()))
        torch.cuda.synchronize()
        res[which] = out.cpu().numpy().reshape(64, 4).copy()
    lane = np.arange(64)
    # 16x16x4: D row = 4*(lane>>4)+reg, col = lane&15
    assert np.array_equal(res[5], (1 + 4 * (lane[:, None] >> 4) + np.arange(4)[None, :]) * 33825.0)
    assert np.array_equal(res[6], np.repeat((1 + (lane & 15))[:, None], 4, 1) * 33825.0)
    # 4x4x1 (16 blocks): block = lane>>2; D[i=reg][j=lane&3] = A[i]*B[j]
    assert np.array_equal(res[1], (4 * (lane[:, None] >> 2) + np.arange(4)[None, :] + 1).astype(np.float32))
    assert np.array_equal(res[2], np.repeat((lane + 1)[:, None], 4, 1).astype(np.float32))
    # cbsz=4, abid=3: every block uses A of block 3 (lanes 12..15)
    assert np.array_equal(res[3], np.repeat((12 + np.arange(4) + 1)[None, :], 64, 0).astype(np.float32))
    # cbsz=2, abid=1: blocks 4g..4g+3 use A of block 4g+1
    assert np.array_equal(res[4], (4 * (4 * (lane[:, None] >> 4) + 1) + np.arange(4)[None, :] + 1).astype(np.float32))


TILED_CASES = [
    # ndim, N, S, cin, cout  (k3 s1; shapes chosen to hit partial tiles, segment seams and image borders)
    (3, 1, (16, 8, 32), 8, 8),
    (3, 2, (16, 24, 64), 8, 8),
    (3, 1, (19, 13, 45), 16, 8),      # ragged: partial tiles in y and x, odd z
    (3, 1, (32, 16, 32), 8, 16),
    (3, 1, (16, 16, 48), 16, 16),
    (2, 2, (16, 256), 16, 16),
    (2, 1, (37, 300), 16, 16),        # ragged 2-D
    (2, 1, (16, 512), 32, 16),
    (2, 1, (24, 256), 8, 8),
]


@pytest.mark.parametrize("case", TILED_CASES)
def test_tiled_conv_fwd_dgrad(case):
    """Family-I kernels (v_mfma_f32_4x4x1_16b, LDS plane ring) forced with algo=3."""
    ndim, N, S, ci, co = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = _rand(rng, (N,) + S + (ci,))
    w = _rand(rng, (3,) * ndim + (ci, co)) * 0.2
    y = O.conv_fwd(x, w, 1)
    dy = _rand(rng, y.shape)
    dx, dw = O.conv_bwd(x, w, 1, dy)
    d = desc(ndim, N, S, ci, co, 3, 1, algo=3)
    xg, wg, dyg = dev(x), dev(w), dev(dy)
    assert rel_err(conv_forward(d, xg, wg, y.shape).cpu().numpy(), y) < TOL
    assert rel_err(conv_backward_data(d, dyg, wg, x.shape).cpu().numpy(), dx) < TOL
    base = torch.ones(x.shape, dtype=torch.float32, device="cuda")
    dxa = conv_backward_data(d, dyg, wg, x.shape, accumulate=1, dx_init=base)
    assert rel_err(dxa.cpu().numpy(), dx + 1.0) < TOL
    if not (ndim == 2 and ci == 32):   # 2-D 32->16 weight gradient exceeds the LDS budget of the tiled kernel
        dwg = conv_backward_weight(d, xg, dyg, w.shape)
        assert rel_err(dwg.cpu().numpy(), dw) < 5e-5
        dwa = conv_backward_weight(d, xg, dyg, w.shape, dw_init=dwg)
        assert rel_err(dwa.cpu().numpy(), 2 * dw) < 5e-5


@pytest.mark.parametrize("case", [(3, 2, (16, 16, 16), 8, 8), (3, 1, (19, 13, 45), 16, 8), (3, 2, (8, 12, 16), 8, 16),
                                  (2, 1, (37, 300), 16, 16), (2, 2, (16, 24), 16, 16)])
def test_conv_forward_fused_statistics(case):
    """BN statistics produced by the conv epilogue (tiled shapes) / the reduction kernel (others)."""
    ndim, N, S, ci, co = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = _rand(rng, (N,) + S + (ci,)) + 0.7
    w = _rand(rng, (3,) * ndim + (ci, co)) * 0.2
    y = O.conv_fwd(x, w, 1)
    ax = tuple(range(y.ndim - 1))
    mu, var = y.mean(axis=ax), y.var(axis=ax)
    lib = _lib.load()
    d = desc(ndim, N, S, ci, co, 3, 1)
    xg, wg = dev(x), dev(w)
    yg = torch.empty(y.shape, dtype=torch.float32, device="cuda")
    mg, rg = torch.empty(co, device="cuda"), torch.empty(co, device="cuda")
    nb = 1 << 22
    scratch = torch.empty(nb, dtype=torch.uint8, device="cuda")
    _lib.check(lib.ursn_conv_forward_stats(ctypes.byref(d), P(xg), P(wg), P(yg), P(mg), P(rg), 1e-3, P(scratch), nb,
                                           stream()))
    torch.cuda.synchronize()
    assert rel_err(yg.cpu().numpy(), y) < TOL
    assert np.abs(mg.cpu().numpy() - mu).max() < 2e-6 * np.sqrt(var.max())  + 1e-6 * np.abs(mu).max()
    assert rel_err(rg.cpu().numpy(), 1 / np.sqrt(var + 1e-3)) < 1e-5


@pytest.mark.parametrize("ndim,S,ci,co", [(3, (16, 16, 32), 8, 3), (2, (16, 256), 16, 3), (2, (16, 256), 16, 5)])
def test_tiled_conv_padded_classes(ndim, S, ci, co):
    """conv2 (F -> num_class): the logits buffers are padded to 4|8 channels so the tiled kernels apply."""
    rng = np.random.default_rng(co)
    N, pc = 2, (co + 3) // 4 * 4
    x = _rand(rng, (N,) + S + (ci,))
    w = _rand(rng, (3,) * ndim + (ci, co)) * 0.2
    y = O.conv_fwd(x, w, 1)
    dy = _rand(rng, y.shape)
    dx, dw = O.conv_bwd(x, w, 1, dy)
    d = desc(ndim, N, S, ci, co, 3, 1, out_cs=pc, algo=3)
    xg, wg = dev(x), dev(w)
    yg = conv_forward(d, xg, wg, y.shape[:-1] + (pc,)).cpu().numpy()
    assert rel_err(yg[..., :co], y) < TOL and np.all(yg[..., co:] == 0)
    dyp = np.zeros(y.shape[:-1] + (pc,))
    dyp[..., :co] = dy
    dyg = dev(dyp)
    assert rel_err(conv_backward_data(d, dyg, wg, x.shape).cpu().numpy(), dx) < TOL
    assert rel_err(conv_backward_weight(d, xg, dyg, w.shape).cpu().numpy(), dw) < 5e-5


@pytest.mark.parametrize("case", [(3, 2, (24, 48, 48), 32, 16), (3, 2, (24, 24, 48), 32, 32), (3, 1, (48, 48, 48), 64, 32),
                                  (3, 1, (48, 48, 32), 16, 48)])
def test_tiled_conv_channel_blocks(case):
    """Layers wider than the instantiated kernels run as 16x16 channel blocks (accumulating launches)."""
    ndim, N, S, ci, co = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = _rand(rng, (N,) + S + (ci,))
    w = _rand(rng, (3,) * ndim + (ci, co)) * 0.1
    y = O.conv_fwd(x, w, 1)
    dy = _rand(rng, y.shape)
    dx, dw = O.conv_bwd(x, w, 1, dy)
    d = desc(ndim, N, S, ci, co, 3, 1, algo=3)
    xg, wg, dyg = dev(x), dev(w), dev(dy)
    lib = _lib.load()
    yg = torch.empty(y.shape, dtype=torch.float32, device="cuda")
    mg, rg = torch.empty(co, device="cuda"), torch.empty(co, device="cuda")
    nb = 1 << 24
    scratch = torch.empty(nb, dtype=torch.uint8, device="cuda")
    _lib.check(lib.ursn_conv_forward_stats(ctypes.byref(d), P(xg), P(wg), P(yg), P(mg), P(rg), 1e-3, P(scratch), nb,
                                           stream()))
    torch.cuda.synchronize()
    assert rel_err(yg.cpu().numpy(), y) < TOL
    ax = tuple(range(y.ndim - 1))
    assert rel_err(mg.cpu().numpy(), y.mean(axis=ax)) < 1e-4
    assert rel_err(rg.cpu().numpy(), 1 / np.sqrt(y.var(axis=ax) + 1e-3)) < 1e-5
    assert rel_err(conv_backward_data(d, dyg, wg, x.shape).cpu().numpy(), dx) < TOL
    base = torch.ones(x.shape, dtype=torch.float32, device="cuda")
    assert rel_err(conv_backward_data(d, dyg, wg, x.shape, accumulate=1, dx_init=base).cpu().numpy(), dx + 1.0) < TOL
    dwg = conv_backward_weight(d, xg, dyg, w.shape)
    assert rel_err(dwg.cpu().numpy(), dw) < 5e-5
    assert rel_err(conv_backward_weight(d, xg, dyg, w.shape, dw_init=dwg).cpu().numpy(), 2 * dw) < 5e-5


@pytest.mark.parametrize("ndim,S,co", [(3, (16, 16, 32), 8), (3, (19, 13, 45), 8), (2, (24, 300), 16)])
def test_tiled_conv0_single_input_channel(ndim, S, co):
    """conv0 (1 -> F): scalar input fetch in the tiled forward, taps-as-rows in the tiled weight gradient."""
    rng = np.random.default_rng(co + ndim)
    N = 2
    x = _rand(rng, (N,) + S + (1,))
    w = _rand(rng, (3,) * ndim + (1, co)) * 0.3
    y = O.conv_fwd(x, w, 1)
    dy = _rand(rng, y.shape)
    _, dw = O.conv_bwd(x, w, 1, dy)
    d = desc(ndim, N, S, 1, co, 3, 1, algo=3)
    xg, wg, dyg = dev(x), dev(w), dev(dy)
    assert rel_err(conv_forward(d, xg, wg, y.shape).cpu().numpy(), y) < TOL
    assert rel_err(conv_backward_weight(d, xg, dyg, w.shape).cpu().numpy(), dw) < 5e-5


@pytest.mark.parametrize("case", [(3, 2, (8, 16, 32), 16, 8), (3, 1, (9, 11, 37), 16, 8), (3, 1, (8, 16, 32), 32, 16),
                                  (2, 2, (12, 256), 32, 16), (2, 1, (9, 300), 16, 8)])
def test_tiled_transposed_conv_forward(case):
    """slim.conv{2,3}d_transpose k3 s2 on the lane-per-low-res-voxel kernel (algo=3), incl. fused BN statistics."""
    ndim, N, S, ci, co = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = _rand(rng, (N,) + S + (ci,))
    w = _rand(rng, (3,) * ndim + (co, ci)) * 0.2
    y = O.deconv_fwd(x, w)
    d = desc(ndim, N, S, ci, co, 3, 2, transposed=1, algo=3)
    xg, wg = dev(x), dev(w)
    lib = _lib.load()
    yg = torch.full(y.shape, float("nan"), dtype=torch.float32, device="cuda")
    mg, rg = torch.empty(co, device="cuda"), torch.empty(co, device="cuda")
    nb = 1 << 24
    scratch = torch.empty(nb, dtype=torch.uint8, device="cuda")
    _lib.check(lib.ursn_conv_forward_stats(ctypes.byref(d), P(xg), P(wg), P(yg), P(mg), P(rg), 1e-3, P(scratch), nb,
                                           stream()))
    torch.cuda.synchronize()
    assert rel_err(yg.cpu().numpy(), y) < TOL
    ax = tuple(range(y.ndim - 1))
    assert np.abs(mg.cpu().numpy() - y.mean(axis=ax)).max() < 1e-5 * np.sqrt(y.var(axis=ax).max())
    assert rel_err(rg.cpu().numpy(), 1 / np.sqrt(y.var(axis=ax) + 1e-3)) < 1e-5
    assert rel_err(conv_forward(d, xg, wg, y.shape).cpu().numpy(), y) < TOL


@pytest.mark.parametrize("case", [(3, 2, (16, 32, 64), 8, 16), (3, 1, (16, 32, 64), 16, 32), (2, 2, (24, 512), 16, 32),
                                  (3, 1, (18, 22, 74), 8, 16)])
def test_tiled_stride2_conv_data_gradient(case):
    """dx of the k3 stride-2 convs (lib/resnet_module.py:43-51) through the same kernel, overwrite and accumulate."""
    ndim, N, S, ci, co = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = _rand(rng, (N,) + S + (ci,))
    w = _rand(rng, (3,) * ndim + (ci, co)) * 0.2
    y = O.conv_fwd(x, w, 2)
    dy = _rand(rng, y.shape)
    dx, _ = O.conv_bwd(x, w, 2, dy)
    d = desc(ndim, N, S, ci, co, 3, 2, algo=3)
    wg, dyg = dev(w), dev(dy)
    assert rel_err(conv_backward_data(d, dyg, wg, x.shape).cpu().numpy(), dx) < TOL
    base = torch.ones(x.shape, dtype=torch.float32, device="cuda")
    assert rel_err(conv_backward_data(d, dyg, wg, x.shape, accumulate=1, dx_init=base).cpu().numpy(), dx + 1.0) < TOL


IGEMM_CASES = [
    (3, 2, (8, 8, 32), 32, 32), (3, 1, (9, 11, 21), 64, 32), (3, 2, (6, 6, 12), 128, 64), (3, 1, (4, 12, 16), 16, 48),
    (2, 2, (32, 48), 32, 64), (2, 1, (19, 37), 64, 128), (2, 1, (16, 16), 256, 32),
    (3, 2, (24, 48, 48), 32, 32), (2, 4, (160, 160), 32, 32),   # enough boxes for the 32-wide co tiles (all-taps, 8-channel chunks)
    (3, 2, (6, 6, 6), 64, 32), (3, 1, (5, 7, 6), 32, 16), (3, 1, (7, 9, 10), 32, 32), (3, 2, (12, 12, 12), 32, 48),  # small-box variants
    (3, 4, (24, 24, 24), 32, 64),   # 24-wide rows as 2 x 12 with 32-wide co tiles
]


@pytest.mark.parametrize("case", IGEMM_CASES)
def test_igemm_conv_fwd_dgrad_stats(case):
    """LDS-staged implicit-GEMM conv (algo=4): forward + fused BN statistics, data gradient, accumulate."""
    ndim, N, S, ci, co = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = _rand(rng, (N,) + S + (ci,))
    w = _rand(rng, (3,) * ndim + (ci, co)) * 0.1
    y = O.conv_fwd(x, w, 1)
    dy = _rand(rng, y.shape)
    dx, dw = O.conv_bwd(x, w, 1, dy)
    d = desc(ndim, N, S, ci, co, 3, 1, algo=4)
    xg, wg, dyg = dev(x), dev(w), dev(dy)
    lib = _lib.load()
    yg = torch.full(y.shape, float("nan"), dtype=torch.float32, device="cuda")
    mg, rg = torch.empty(co, device="cuda"), torch.empty(co, device="cuda")
    nb = 1 << 24
    scratch = torch.empty(nb, dtype=torch.uint8, device="cuda")
    _lib.check(lib.ursn_conv_forward_stats(ctypes.byref(d), P(xg), P(wg), P(yg), P(mg), P(rg), 1e-3, P(scratch), nb,
                                           stream()))
    torch.cuda.synchronize()
    assert rel_err(yg.cpu().numpy(), y) < TOL
    ax = tuple(range(y.ndim - 1))
    assert np.abs(mg.cpu().numpy() - y.mean(axis=ax)).max() < 1e-5 * np.sqrt(y.var(axis=ax).max())
    assert rel_err(rg.cpu().numpy(), 1 / np.sqrt(y.var(axis=ax) + 1e-3)) < 1e-5
    assert rel_err(conv_forward(d, xg, wg, y.shape).cpu().numpy(), y) < TOL
    assert rel_err(conv_backward_data(d, dyg, wg, x.shape).cpu().numpy(), dx) < TOL
    base = torch.ones(x.shape, dtype=torch.float32, device="cuda")
    assert rel_err(conv_backward_data(d, dyg, wg, x.shape, accumulate=1, dx_init=base).cpu().numpy(), dx + 1.0) < TOL
    dwg = conv_backward_weight(d, xg, dyg, w.shape)
    assert rel_err(dwg.cpu().numpy(), dw) < 5e-5
    assert rel_err(conv_backward_weight(d, xg, dyg, w.shape, dw_init=dwg).cpu().numpy(), 2 * dw) < 5e-5


DEEP_CASES = [
    # N, S, cin, cout (2-D shapes: the 9-tap form, no z halo): the deepest levels of an F = 8 network (conv_deep.hip): 12^3 x 128, 6^3 x 256 (whole-image boxes, the
    # chunks split over workgroups as well), the 2C -> C decoder layers (four rounds of chunks), 8^3, ragged boxes, produced
    # channels that are not a multiple of 64
    (4, (12, 12, 12), 128, 128), (4, (6, 6, 6), 256, 256), (1, (12, 12, 12), 256, 128), (2, (8, 8, 8), 256, 256),
    (1, (5, 7, 9), 128, 64), (2, (6, 6, 6), 128, 256), (1, (16, 16, 16), 128, 128), (1, (4, 6, 10), 192, 80),
    (4, (32, 32), 128, 128), (4, (16, 16), 256, 256), (4, (8, 8), 512, 512), (2, (16, 16), 512, 256), (1, (9, 21), 128, 64),
]


@pytest.mark.parametrize("case", DEEP_CASES)
def test_deep_level_conv_fwd_dgrad_stats(case):
    """Weight-streaming kernel of the deepest levels (lib/resnet_module.py:43-66 at levels 4-5): forward + fused BatchNorm
    statistics, data gradient, accumulate -- and the dispatcher really hands these shapes to it."""
    N, S, ci, co = case
    nd = len(S)
    rng = np.random.default_rng(ci * 7 + co + S[-1])
    x = _rand(rng, (N,) + S + (ci,))
    w = _rand(rng, (3,) * nd + (ci, co)) * 0.05
    y = O.conv_fwd(x, w, 1)
    dy = _rand(rng, y.shape)
    dx, dw = O.conv_bwd(x, w, 1, dy)
    d = desc(nd, N, S, ci, co, 3, 1)
    lib = _lib.load()
    buf = ctypes.create_string_buffer(32)
    for ps, want in ((0, b"dconv"), (1, b"dconv" if co >= 128 and co % 64 == 0 else None)):
        _lib.check(lib.ursn_conv_plan(ctypes.byref(d), ps, buf, 32))
        assert want is None or buf.value == want, (ps, buf.value)
    xg, wg, dyg = dev(x), dev(w), dev(dy)
    yg = torch.full(y.shape, float("nan"), dtype=torch.float32, device="cuda")
    mg, rg = torch.empty(co, device="cuda"), torch.empty(co, device="cuda")
    nb = 1 << 24
    scratch = torch.empty(nb, dtype=torch.uint8, device="cuda")
    _lib.check(lib.ursn_conv_forward_stats(ctypes.byref(d), P(xg), P(wg), P(yg), P(mg), P(rg), 1e-3, P(scratch), nb, stream()))
    torch.cuda.synchronize()
    assert lib.ursn_last_kernel_name().startswith(b"dconv")
    assert rel_err(yg.cpu().numpy(), y) < TOL
    ax = tuple(range(y.ndim - 1))
    assert np.abs(mg.cpu().numpy() - y.mean(axis=ax)).max() < 1e-5 * np.sqrt(y.var(axis=ax).max())
    assert rel_err(rg.cpu().numpy(), 1 / np.sqrt(y.var(axis=ax) + 1e-3)) < 1e-5
    assert rel_err(conv_forward(d, xg, wg, y.shape).cpu().numpy(), y) < TOL
    assert rel_err(conv_backward_data(d, dyg, wg, x.shape).cpu().numpy(), dx) < TOL
    base = _rand(rng, x.shape)
    assert rel_err(conv_backward_data(d, dyg, wg, x.shape, accumulate=1, dx_init=dev(base)).cpu().numpy(), dx + base) < TOL
    # weight gradient: both operands straight from L2 (wgrad_deep.hip) where the level is small enough; accumulates
    _lib.check(lib.ursn_conv_plan(ctypes.byref(d), 2, buf, 32))
    small = N * int(np.prod(S)) * (1 if nd == 3 else 4) <= 1024   # only the last level (wgrad_deep.hip: measured)
    assert (buf.value == b"dwgrad") == small, buf.value
    dwg = conv_backward_weight(d, xg, dyg, w.shape)
    assert lib.ursn_last_kernel_name().startswith(b"dwgrad") == small
    assert rel_err(dwg.cpu().numpy(), dw) < 5e-5
    assert rel_err(conv_backward_weight(d, xg, dyg, w.shape, dw_init=dwg).cpu().numpy(), 2 * dw) < 5e-5


@pytest.mark.parametrize("case", [(3, 2, (8, 12, 16), 16, 8, 1), (3, 2, (8, 12, 16), 8, 16, 2), (3, 1, (6, 10, 14), 32, 16, 1),
                                  (3, 1, (8, 8, 12), 16, 32, 2), (2, 2, (16, 40), 32, 16, 1), (2, 2, (16, 40), 32, 64, 2),
                                  (3, 1, (4, 6, 10), 64, 32, 1)])
def test_pointwise_shortcut_conv(case):
    """1x1 shortcut convs (lib/resnet_module.py:25-33), stride 1 and 2, on the dedicated kernels (algo=5)."""
    ndim, N, S, ci, co, st = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = _rand(rng, (N,) + S + (ci,))
    w = _rand(rng, (1,) * ndim + (ci, co)) * 0.3
    y = O.conv_fwd(x, w, st)
    dy = _rand(rng, y.shape)
    dx, dw = O.conv_bwd(x, w, st, dy)
    d = desc(ndim, N, S, ci, co, 1, st, algo=5)
    xg, wg, dyg = dev(x), dev(w), dev(dy)
    lib = _lib.load()
    yg = torch.full(y.shape, float("nan"), dtype=torch.float32, device="cuda")
    mg, rg = torch.empty(co, device="cuda"), torch.empty(co, device="cuda")
    nb = 1 << 22
    scratch = torch.empty(nb, dtype=torch.uint8, device="cuda")
    _lib.check(lib.ursn_conv_forward_stats(ctypes.byref(d), P(xg), P(wg), P(yg), P(mg), P(rg), 1e-3, P(scratch), nb,
                                           stream()))
    torch.cuda.synchronize()
    assert rel_err(yg.cpu().numpy(), y) < TOL
    ax = tuple(range(y.ndim - 1))
    assert np.abs(mg.cpu().numpy() - y.mean(axis=ax)).max() < 1e-5 * np.sqrt(y.var(axis=ax).max())
    assert rel_err(rg.cpu().numpy(), 1 / np.sqrt(y.var(axis=ax) + 1e-3)) < 1e-5
    base = torch.ones(x.shape, dtype=torch.float32, device="cuda")
    assert rel_err(conv_backward_data(d, dyg, wg, x.shape, accumulate=1, dx_init=base).cpu().numpy(), dx + 1.0) < TOL
    if st == 1:
        assert rel_err(conv_backward_data(d, dyg, wg, x.shape).cpu().numpy(), dx) < TOL
    dwg = conv_backward_weight(d, xg, dyg, w.shape)
    assert rel_err(dwg.cpu().numpy(), dw) < 5e-5


S2_CASES = [(3, 2, (16, 32, 64), 8, 16), (3, 1, (17, 21, 45), 16, 32), (3, 1, (8, 16, 32), 32, 64), (2, 2, (48, 160), 16, 32),
            (2, 1, (37, 75), 8, 16), (3, 1, (6, 6, 6), 64, 128)]


@pytest.mark.parametrize("case", S2_CASES)
def test_stride2_lds_conv_forward_and_weight_gradient(case):
    """k3 stride-2 convs (lib/resnet_module.py:43-51) on the LDS-staged stride-2 kernels (algo=6): forward with fused
    BN statistics, weight gradient (overwrite + accumulate); odd sizes exercise TF SAME pad-before = 1."""
    ndim, N, S, ci, co = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = _rand(rng, (N,) + S + (ci,))
    w = _rand(rng, (3,) * ndim + (ci, co)) * 0.2
    y = O.conv_fwd(x, w, 2)
    dy = _rand(rng, y.shape)
    _, dw = O.conv_bwd(x, w, 2, dy)
    d = desc(ndim, N, S, ci, co, 3, 2, algo=6)
    xg, wg, dyg = dev(x), dev(w), dev(dy)
    lib = _lib.load()
    yg = torch.full(y.shape, float("nan"), dtype=torch.float32, device="cuda")
    mg, rg = torch.empty(co, device="cuda"), torch.empty(co, device="cuda")
    nb = 1 << 24
    scratch = torch.empty(nb, dtype=torch.uint8, device="cuda")
    _lib.check(lib.ursn_conv_forward_stats(ctypes.byref(d), P(xg), P(wg), P(yg), P(mg), P(rg), 1e-3, P(scratch), nb,
                                           stream()))
    torch.cuda.synchronize()
    assert rel_err(yg.cpu().numpy(), y) < TOL
    ax = tuple(range(y.ndim - 1))
    assert np.abs(mg.cpu().numpy() - y.mean(axis=ax)).max() < 1e-5 * np.sqrt(y.var(axis=ax).max())
    assert rel_err(rg.cpu().numpy(), 1 / np.sqrt(y.var(axis=ax) + 1e-3)) < 1e-5
    assert rel_err(conv_forward(d, xg, wg, y.shape).cpu().numpy(), y) < TOL
    dwg = conv_backward_weight(d, xg, dyg, w.shape)
    assert rel_err(dwg.cpu().numpy(), dw) < 5e-5
    assert rel_err(conv_backward_weight(d, xg, dyg, w.shape, dw_init=dwg).cpu().numpy(), 2 * dw) < 5e-5


@pytest.mark.parametrize("case", [(3, 2, (8, 16, 32), 16, 8), (3, 1, (9, 11, 37), 32, 16), (2, 2, (24, 80), 32, 16),
                                  (2, 1, (19, 41), 16, 8)])
def test_stride2_lds_transposed_conv_gradients(case):
    """dx and dW of slim.conv{2,3}d_transpose k3 s2 (lib/uresnet.py:72-79) on the LDS-staged stride-2 kernels (algo=6)."""
    ndim, N, S, ci, co = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = _rand(rng, (N,) + S + (ci,))
    w = _rand(rng, (3,) * ndim + (co, ci)) * 0.2
    y = O.deconv_fwd(x, w)
    dy = _rand(rng, y.shape)
    dx, dw = O.deconv_bwd(x, w, dy)
    d = desc(ndim, N, S, ci, co, 3, 2, transposed=1, algo=6)
    xg, wg, dyg = dev(x), dev(w), dev(dy)
    assert rel_err(conv_backward_data(d, dyg, wg, x.shape).cpu().numpy(), dx) < TOL
    base = torch.ones(x.shape, dtype=torch.float32, device="cuda")
    assert rel_err(conv_backward_data(d, dyg, wg, x.shape, accumulate=1, dx_init=base).cpu().numpy(), dx + 1.0) < TOL
    dwg = conv_backward_weight(d, xg, dyg, w.shape)
    assert rel_err(dwg.cpu().numpy(), dw) < 5e-5


@pytest.mark.parametrize("ndim,S,k", [(3, (8, 16, 32), 3), (3, (9, 11, 37), 3), (3, (8, 16, 32), 1), (3, (5, 7, 19), 1),
                                      (2, (16, 256), 3), (2, (12, 40), 1)])
def test_split_input_concat_never_materialised(ndim, S, k):
    """tf.concat([deconv, skip], axis=-1) feeding resnet_module (lib/uresnet.py:81-84) with the two halves kept as
    separate tensors (in_split / x2 / dx2): forward + fused statistics, data gradient into both halves (overwrite and
    accumulate), weight gradient -- against the oracle on the concatenated input."""
    N, h, co = 2, 8, 8
    rng = np.random.default_rng(7 * ndim + k)
    xa, xb = _rand(rng, (N,) + S + (h,)), _rand(rng, (N,) + S + (h,))
    x = np.concatenate([xa, xb], axis=-1)
    w = _rand(rng, (k,) * ndim + (2 * h, co)) * 0.2
    y = O.conv_fwd(x, w, 1)
    dy = _rand(rng, y.shape)
    dx, dw = O.conv_bwd(x, w, 1, dy)
    xag, xbg, wg, dyg = dev(xa), dev(xb), dev(w), dev(dy)
    d = desc(ndim, N, S, 2 * h, co, k, 1)
    d.in_split, d.in_cstride, d.in2_cstride = h, h, h
    d.x2 = xbg.data_ptr()
    lib = _lib.load()
    assert rel_err(conv_forward(d, xag, wg, y.shape).cpu().numpy(), y) < TOL
    yg = torch.full(y.shape, float("nan"), dtype=torch.float32, device="cuda")
    mg, rg = torch.empty(co, device="cuda"), torch.empty(co, device="cuda")
    nb = 1 << 24
    scratch = torch.empty(nb, dtype=torch.uint8, device="cuda")
    _lib.check(lib.ursn_conv_forward_stats(ctypes.byref(d), P(xag), P(wg), P(yg), P(mg), P(rg), 1e-3, P(scratch), nb,
                                           stream()))
    torch.cuda.synchronize()
    assert rel_err(yg.cpu().numpy(), y) < TOL
    ax = tuple(range(y.ndim - 1))
    assert np.abs(mg.cpu().numpy() - y.mean(axis=ax)).max() < 1e-5 * np.sqrt(y.var(axis=ax).max())
    assert rel_err(rg.cpu().numpy(), 1 / np.sqrt(y.var(axis=ax) + 1e-3)) < 1e-5
    for acc in (0, 1):
        dxa = torch.full(xa.shape, 1.0 if acc else float("nan"), dtype=torch.float32, device="cuda")
        dxb = torch.full(xb.shape, 1.0 if acc else float("nan"), dtype=torch.float32, device="cuda")
        d.dx2 = dxb.data_ptr()
        _lib.check(lib.ursn_conv_backward_data(ctypes.byref(d), P(dyg), P(wg), P(dxa), acc, stream()))
        torch.cuda.synchronize()
        assert rel_err(dxa.cpu().numpy(), dx[..., :h] + acc) < TOL
        assert rel_err(dxb.cpu().numpy(), dx[..., h:] + acc) < TOL
    dwg = conv_backward_weight(d, xag, dyg, w.shape)
    assert rel_err(dwg.cpu().numpy(), dw) < 5e-5
    if k == 3:  # a split no two-tensor kernel covers must fail loudly, not fall back
        d2 = desc(ndim, N, S, 2 * h, co, k, 1)
        d2.in_split, d2.x2 = 4, xbg.data_ptr()
        yy = torch.empty(y.shape, dtype=torch.float32, device="cuda")
        assert lib.ursn_conv_forward(ctypes.byref(d2), P(xag), P(wg), P(yy), stream()) != 0


@pytest.mark.parametrize("ndim,S,ci,co,split", [(3, (8, 16, 32), 16, 8, True), (3, (9, 11, 37), 16, 8, False),
                                                (3, (8, 16, 32), 8, 8, False), (2, (16, 256), 16, 16, False)])
def test_tiled_dgrad_with_fused_shortcut_term(ndim, S, ci, co, split):
    """dx of resnet_conv1 and of the parallel 1x1 shortcut (lib/resnet_module.py:25-43) in ONE kernel (pw_dy / pw_w)."""
    N = 2
    rng = np.random.default_rng(5 * ndim + ci + co)
    x = _rand(rng, (N,) + S + (ci,))
    w = _rand(rng, (3,) * ndim + (ci, co)) * 0.2
    ws = _rand(rng, (1,) * ndim + (ci, co)) * 0.3
    dy, dys = _rand(rng, (N,) + S + (co,)), _rand(rng, (N,) + S + (co,))
    dx = O.conv_bwd(x, w, 1, dy)[0] + O.conv_bwd(x, ws, 1, dys)[0]
    wg, wsg, dyg, dysg = dev(w), dev(ws), dev(dy), dev(dys)
    d = desc(ndim, N, S, ci, co, 3, 1)
    d.pw_dy, d.pw_w = dysg.data_ptr(), wsg.data_ptr()
    lib = _lib.load()
    h = ci // 2
    for acc in (0, 1):
        if split:
            d.in_split, d.in_cstride, d.in2_cstride = h, h, h
            dxa = torch.full((N,) + S + (h,), 1.0 if acc else float("nan"), dtype=torch.float32, device="cuda")
            dxb = torch.full((N,) + S + (h,), 1.0 if acc else float("nan"), dtype=torch.float32, device="cuda")
            d.dx2 = dxb.data_ptr()
            _lib.check(lib.ursn_conv_backward_data(ctypes.byref(d), P(dyg), P(wg), P(dxa), acc, stream()))
            torch.cuda.synchronize()
            got = np.concatenate([dxa.cpu().numpy(), dxb.cpu().numpy()], axis=-1)
        else:
            base = torch.full(x.shape, 1.0 if acc else float("nan"), dtype=torch.float32, device="cuda")
            got = conv_backward_data(d, dyg, wg, x.shape, accumulate=acc, dx_init=base).cpu().numpy()
        assert rel_err(got, dx + acc) < TOL
    # the forward pass has no such term: must be refused
    yy = torch.empty(dy.shape, dtype=torch.float32, device="cuda")
    assert lib.ursn_conv_forward(ctypes.byref(d), P(dev(x)), P(wg), P(yy), stream()) != 0


@pytest.mark.parametrize("S,ci,co,cs", [((16, 32, 64), 8, 16, 0), ((18, 22, 74), 8, 16, 0), ((16, 16, 64), 8, 16, 24), ((16, 32, 64), 16, 16, 0)])
def test_stride2_dgrad_with_fused_stride2_shortcut_term(S, ci, co, cs):
    """dx of a unit's stride-2 resnet_conv1 AND of its 1x1 stride-2 shortcut (lib/resnet_module.py:25-43) in one launch of the
    lane-per-low-res-voxel kernel (tdeconv + pw): the shortcut touches the even-even-even voxels only.  Overwrite and accumulate;
    cs: channel stride of the shortcut's gradient tensor (0 = compact)."""
    N, ndim = 2, 3
    rng = np.random.default_rng(S[0] + S[2] + ci + cs)
    x = _rand(rng, (N,) + S + (ci,))
    w = _rand(rng, (3,) * ndim + (ci, co)) * 0.2
    ws = _rand(rng, (1,) * ndim + (ci, co)) * 0.3
    Slo = tuple((v + 1) // 2 for v in S)
    dy, dys = _rand(rng, (N,) + Slo + (co,)), _rand(rng, (N,) + Slo + (co,))
    dx = O.conv_bwd(x, w, 2, dy)[0] + O.conv_bwd(x, ws, 2, dys)[0]
    wg, wsg, dyg = dev(w), dev(ws), dev(dy)
    if cs:
        wide = np.full((N,) + Slo + (cs,), np.nan, dtype=np.float32)
        wide[..., :co] = dys
        dysg = dev(wide)
    else:
        dysg = dev(dys)
    d = desc(ndim, N, S, ci, co, 3, 2)
    d.pw_dy, d.pw_w, d.pw_dy_cstride = dysg.data_ptr(), wsg.data_ptr(), cs
    lib = _lib.load()
    lib.ursn_last_kernel_name.restype = ctypes.c_char_p
    for acc in (0, 1):
        base = torch.full(x.shape, 1.0 if acc else float("nan"), dtype=torch.float32, device="cuda")
        got = conv_backward_data(d, dyg, wg, x.shape, accumulate=acc, dx_init=base).cpu().numpy()
        name = lib.ursn_last_kernel_name().decode()
        assert name.startswith("tdeconv<16,") and name.endswith("+pw"), name
        assert rel_err(got, dx + acc) < TOL


@pytest.mark.parametrize("case", [(3, 2, (8, 16, 32), 32, 16), (3, 1, (5, 7, 19), 16, 16), (3, 1, (6, 6, 6), 64, 32),
                                  (2, 2, (24, 40), 32, 16), (2, 1, (17, 35), 16, 32)])
def test_lds_scatter_transposed_conv_forward(case):
    """slim.conv{2,3}d_transpose k3 s2 (lib/uresnet.py:72-79) on the LDS-staged scatter kernel (algo=7) + fused BN stats."""
    ndim, N, S, ci, co = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = _rand(rng, (N,) + S + (ci,))
    w = _rand(rng, (3,) * ndim + (co, ci)) * 0.2
    y = O.deconv_fwd(x, w)
    d = desc(ndim, N, S, ci, co, 3, 2, transposed=1, algo=7)
    xg, wg = dev(x), dev(w)
    lib = _lib.load()
    yg = torch.full(y.shape, float("nan"), dtype=torch.float32, device="cuda")
    mg, rg = torch.empty(co, device="cuda"), torch.empty(co, device="cuda")
    nb = 1 << 24
    scratch = torch.empty(nb, dtype=torch.uint8, device="cuda")
    _lib.check(lib.ursn_conv_forward_stats(ctypes.byref(d), P(xg), P(wg), P(yg), P(mg), P(rg), 1e-3, P(scratch), nb,
                                           stream()))
    torch.cuda.synchronize()
    assert rel_err(yg.cpu().numpy(), y) < TOL
    ax = tuple(range(y.ndim - 1))
    assert np.abs(mg.cpu().numpy() - y.mean(axis=ax)).max() < 1e-5 * np.sqrt(y.var(axis=ax).max())
    assert rel_err(rg.cpu().numpy(), 1 / np.sqrt(y.var(axis=ax) + 1e-3)) < 1e-5
    assert rel_err(conv_forward(d, xg, wg, y.shape).cpu().numpy(), y) < TOL


@pytest.mark.parametrize("case", [(3, 2, (16, 32, 64), 16, 32), (3, 1, (10, 14, 38), 16, 16), (2, 2, (24, 96), 16, 32),
                                  (3, 1, (12, 12, 12), 32, 64)])
def test_lds_scatter_stride2_conv_data_gradient(case):
    """dx of the k3 stride-2 convs (lib/resnet_module.py:43-51) on the LDS-staged scatter kernel, overwrite + accumulate."""
    ndim, N, S, ci, co = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = _rand(rng, (N,) + S + (ci,))
    w = _rand(rng, (3,) * ndim + (ci, co)) * 0.2
    y = O.conv_fwd(x, w, 2)
    dy = _rand(rng, y.shape)
    dx, _ = O.conv_bwd(x, w, 2, dy)
    d = desc(ndim, N, S, ci, co, 3, 2, algo=7)
    wg, dyg = dev(w), dev(dy)
    assert rel_err(conv_backward_data(d, dyg, wg, x.shape).cpu().numpy(), dx) < TOL
    base = torch.ones(x.shape, dtype=torch.float32, device="cuda")
    assert rel_err(conv_backward_data(d, dyg, wg, x.shape, accumulate=1, dx_init=base).cpu().numpy(), dx + 1.0) < TOL


@pytest.mark.parametrize("ndim,S,C", [(3, (8, 16, 32), 8), (3, (9, 11, 37), 16), (3, (16, 16, 32), 16), (2, (16, 256), 16),
                                      (2, (12, 300), 8)])
def test_normalise_on_load_conv_and_weight_gradient(ndim, S, C):
    """resnet_conv2 reading the RAW output of resnet_conv1 (lib/resnet_module.py:43-51: BatchNorm without activation in
    between): the kernels apply (z - mean) * rstd + beta while staging (in_mean / in_rstd / in_beta), padding stays zero."""
    N = 2
    rng = np.random.default_rng(3 * ndim + C)
    z = _rand(rng, (N,) + S + (C,)) * 2.0 + 0.5
    mean, rstd, beta = _rand(rng, (C,)) * 0.3, 0.5 + rng.random(C), _rand(rng, (C,)) * 0.2
    x = (z - mean) * rstd + beta
    w = _rand(rng, (3,) * ndim + (C, C)) * 0.2
    y = O.conv_fwd(x, w, 1)
    dy = _rand(rng, y.shape)
    _, dw = O.conv_bwd(x, w, 1, dy)
    zg, wg, dyg = dev(z), dev(w), dev(dy)
    mg_, rg_, bg_ = dev(mean), dev(rstd), dev(beta)
    d = desc(ndim, N, S, C, C, 3, 1)
    d.in_mean, d.in_rstd, d.in_beta = mg_.data_ptr(), rg_.data_ptr(), bg_.data_ptr()
    lib = _lib.load()
    assert rel_err(conv_forward(d, zg, wg, y.shape).cpu().numpy(), y) < TOL
    yg = torch.full(y.shape, float("nan"), dtype=torch.float32, device="cuda")
    mo, ro = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    nb = 1 << 24
    scratch = torch.empty(nb, dtype=torch.uint8, device="cuda")
    _lib.check(lib.ursn_conv_forward_stats(ctypes.byref(d), P(zg), P(wg), P(yg), P(mo), P(ro), 1e-3, P(scratch), nb,
                                           stream()))
    torch.cuda.synchronize()
    assert rel_err(yg.cpu().numpy(), y) < TOL
    ax = tuple(range(y.ndim - 1))
    assert np.abs(mo.cpu().numpy() - y.mean(axis=ax)).max() < 1e-5 * np.sqrt(y.var(axis=ax).max())
    dwg = conv_backward_weight(d, zg, dyg, w.shape)
    assert rel_err(dwg.cpu().numpy(), dw) < 5e-5
    # shapes without a staging-affine kernel must be refused
    d2 = desc(ndim, N, S, C, 32, 3, 1)
    d2.in_mean, d2.in_rstd, d2.in_beta = mg_.data_ptr(), rg_.data_ptr(), bg_.data_ptr()
    w2 = dev(_rand(rng, (3,) * ndim + (C, 32)))
    yy = torch.empty((N,) + S + (32,), dtype=torch.float32, device="cuda")
    if not (ndim == 2 and C == 16):   # 2-D 16 -> 32 is a native tiled shape
        assert lib.ursn_conv_forward(ctypes.byref(d2), P(zg), P(w2), P(yy), stream()) != 0


@pytest.mark.parametrize("ndim,N,S,ci,co", [(3, 2, (8, 8, 32), 64, 32), (3, 1, (6, 6, 6), 64, 32), (3, 2, (12, 12, 12), 32, 16),
                                            (3, 2, (24, 48, 48), 64, 32), (2, 2, (32, 48), 32, 16), (2, 4, (160, 160), 64, 32)])
def test_igemm_dgrad_with_fused_shortcut_term(ndim, N, S, ci, co):
    """Decoder units (2C -> C): dx of resnet_conv1 and of the parallel 1x1 shortcut (lib/resnet_module.py:25-43) in the
    all-taps implicit-GEMM data-gradient kernel (pw_dy / pw_w), standard and small-box variants, 16- and 32-wide tiles."""
    rng = np.random.default_rng(11 * ndim + ci + co + S[-1])
    x = _rand(rng, (N,) + S + (ci,))
    w = _rand(rng, (3,) * ndim + (ci, co)) * 0.1
    ws = _rand(rng, (1,) * ndim + (ci, co)) * 0.3
    dy, dys = _rand(rng, (N,) + S + (co,)), _rand(rng, (N,) + S + (co,))
    dx = O.conv_bwd(x, w, 1, dy)[0] + O.conv_bwd(x, ws, 1, dys)[0]
    wg, wsg, dyg, dysg = dev(w), dev(ws), dev(dy), dev(dys)
    d = desc(ndim, N, S, ci, co, 3, 1)
    d.pw_dy, d.pw_w = dysg.data_ptr(), wsg.data_ptr()
    for acc in (0, 1):
        base = torch.full(x.shape, 1.0 if acc else float("nan"), dtype=torch.float32, device="cuda")
        got = conv_backward_data(d, dyg, wg, x.shape, accumulate=acc, dx_init=base).cpu().numpy()
        assert rel_err(got, dx + acc) < TOL


# ---- numerics hardening: fused one-pass statistics under |mean| >> std ------------------------------------------------
# TensorFlow's moments are two-pass; the conv epilogues accumulate sum / sum of squares in one
# pass.  A conv whose output carries a large per-channel offset (z = offset + N(0,1)) is the regime where a naive
# one-pass variance cancels catastrophically; every kernel family with fused statistics is driven through it here
# against the two-pass fp64 oracle.
OFFSET_CASES = [
    # tag, ndim, S, cin, cout, k, stride, transposed, algo
    ("tconv8", 3, (32, 32, 64), 8, 8, 3, 1, 0, 3),
    ("tconv16", 3, (16, 32, 64), 16, 16, 3, 1, 0, 3),
    ("tconv2d16", 2, (64, 512), 16, 16, 3, 1, 0, 3),
    ("igemm32", 3, (16, 24, 48), 32, 32, 3, 1, 0, 4),
    ("igemm64", 3, (12, 12, 24), 64, 64, 3, 1, 0, 4),
    ("igemm2d64", 2, (64, 128), 64, 64, 3, 1, 0, 4),
    ("pconv16_8", 3, (16, 32, 64), 16, 8, 1, 1, 0, 5),
    ("pconv_s2", 3, (16, 32, 64), 8, 16, 1, 2, 0, 5),
    ("s2conv", 3, (16, 32, 64), 8, 16, 3, 2, 0, 6),
    ("tdeconv16_8", 3, (8, 16, 32), 16, 8, 3, 2, 1, 3),
    ("s2scatter32_16", 3, (8, 16, 32), 32, 16, 3, 2, 1, 7),
    ("generic", 3, (8, 16, 32), 12, 20, 3, 1, 0, 2),
]


@pytest.mark.parametrize("ratio", [1e2, 1e3])
@pytest.mark.parametrize("case", OFFSET_CASES, ids=[c[0] for c in OFFSET_CASES])
def test_fused_statistics_with_large_mean_to_std_ratio(case, ratio):
    tag, ndim, S, ci, co, k, st, tr, algo = case
    rng = np.random.default_rng(len(tag) + int(ratio))
    N = 2
    x = _rand(rng, (N,) + S + (ci,))
    # weights that copy input channel (co % ci) of ONE tap (conv) / of the 2^d taps that tile the output (transposed conv):
    # z is exactly an input sample, so z = offset + N(0,1) at every output voxel (no border effect)
    if tr:
        w = np.zeros((k,) * ndim + (co, ci))
        for t in np.ndindex(*((2,) * ndim)):
            for c in range(co):
                w[t + (c, c % ci)] = 1.0
    else:
        w = np.zeros((k,) * ndim + (ci, co))
        tap = (0,) * ndim if st == 2 or k == 1 else (1,) * ndim
        for c in range(co):
            w[tap + (c % ci, c)] = 1.0
    off = ratio * (1.0 + 0.25 * np.arange(ci) / ci) * np.where(np.arange(ci) % 2, -1.0, 1.0)
    x = (x + off).astype(np.float32).astype(np.float64)
    y = O.deconv_fwd(x, w) if tr else O.conv_fwd(x, w, st)
    ax = tuple(range(y.ndim - 1))
    mu = y.mean(axis=ax)
    var = ((y - mu) ** 2).mean(axis=ax)          # two-pass
    assert np.all(np.abs(mu) / np.sqrt(var) > 0.9 * ratio)
    d = desc(ndim, N, S, ci, co, k, st, transposed=tr, algo=algo)
    lib = _lib.load()
    xg, wg = dev(x), dev(w)
    yg = torch.empty(y.shape, dtype=torch.float32, device="cuda")
    mg, rg = torch.empty(co, device="cuda"), torch.empty(co, device="cuda")
    nb = 1 << 24
    scratch = torch.empty(nb, dtype=torch.uint8, device="cuda")
    _lib.check(lib.ursn_conv_forward_stats(ctypes.byref(d), P(xg), P(wg), P(yg), P(mg), P(rg), 1e-3, P(scratch), nb,
                                           stream()))
    torch.cuda.synchronize()
    assert np.array_equal(yg.cpu().numpy(), y.astype(np.float32))
    e_mu = float((np.abs(mg.cpu().numpy() - mu) / np.sqrt(var)).max())
    e_rs = rel_err(rg.cpu().numpy(), 1 / np.sqrt(var + 1e-3))
    print("%s |mean|/std %g: mean error %.2e std, rstd rel error %.2e" % (tag, ratio, e_mu, e_rs))
    # the fp32 output itself is exact here; what is left is the rounding of the fp32 mean (ulp(mean)/std)
    assert e_mu < 1e-7 * ratio + 1e-5
    assert e_rs < 1e-4


def _join_mask_words(keep):
    """The bit mask a residual join's forward pass writes for an 8-channel tensor: every 32 voxels (256 elements) take four
    64-bit words; channel 4*q+j of voxel v is bit (v % 32) * 2 + q of word (v // 32) * 4 + j (include/uresnet_hip.h)."""
    k = keep.reshape(-1, 8)
    V = k.shape[0]
    words = np.zeros(((V + 31) // 32) * 4, dtype=np.uint64)
    for q in range(2):
        for j in range(4):
            idx = np.nonzero(k[:, 4 * q + j])[0]
            np.bitwise_or.at(words, (idx // 32) * 4 + j, np.uint64(1) << ((idx % 32) * 2 + q).astype(np.uint64))
    return words


@pytest.mark.parametrize("co,split,relu,two", [(8, False, 0, False), (8, False, 1, False), (8, False, 2, True),
                                               (8, False, 2, False), (3, False, 1, False), (8, True, 1, False)])
@pytest.mark.parametrize("S", [(8, 16, 32), (9, 11, 37)])
def test_data_gradient_with_fused_bn_backward_reductions(S, co, split, relu, two):
    """The data gradient of a 3^3 conv is the LAST contribution to d(input); the BatchNorm backward of the layer(s) that
    produced that input starts with sum g, sum g*xhat(, sum g*xhat2), g = dx * relu mask (lib/resnet_module.py:43-60 under
    tf.gradients).  ursn_conv_desc.bs_* takes those sums in the data-gradient kernel's epilogue."""
    N, ci = 2, 16 if split else 8
    rng = np.random.default_rng(S[2] + co + 7 * relu + two)
    x_shape = (N,) + S + (ci,)
    w = _rand(rng, (3, 3, 3, ci, co)) * 0.2
    dy = _rand(rng, (N,) + S + (co,))
    dx = O.conv_bwd(np.zeros(x_shape), w, 1, dy)[0]
    z, z2 = _rand(rng, (N,) + S + (8,)) * 1.5 + 0.3, _rand(rng, (N,) + S + (8,)) * 0.7 - 0.2
    ax = (0, 1, 2, 3)
    mean, rstd = z.mean(axis=ax), 1.0 / np.sqrt(z.var(axis=ax) + 1e-3)
    mean2, rstd2 = z2.mean(axis=ax), 1.0 / np.sqrt(z2.var(axis=ax) + 1e-3)
    beta = _rand(rng, (8,)) * 0.3
    f32 = lambda a: np.asarray(a, dtype=np.float32)   # noqa: E731
    if relu == 1:   # the kernel's own expression in fp32, so that the two sides agree on borderline elements
        keep = (f32(z) * f32(rstd) + (f32(beta) - f32(mean) * f32(rstd))) > 0
    elif relu == 2:
        keep = rng.random(z.shape) < 0.6
    else:
        keep = np.ones(z.shape, dtype=bool)
    pc = (co + 3) // 4 * 4   # the logits layer's gradient buffer is padded to 4 channels
    dyp = np.zeros(dy.shape[:-1] + (pc,))
    dyp[..., :co] = dy
    wg, dyg, zg, z2g = dev(w), dev(dyp), dev(z), dev(z2)
    mg_, rg_, bg_, m2g, r2g = dev(mean), dev(rstd), dev(beta), dev(mean2), dev(rstd2)
    maskg = torch.from_numpy(_join_mask_words(keep).view(np.int64)).cuda()
    lib = _lib.load()
    d = desc(3, N, S, ci, co, 3, 1, out_cs=pc)
    d.bs_z, d.bs_mean, d.bs_rstd, d.bs_beta, d.bs_z_cstride = zg.data_ptr(), mg_.data_ptr(), rg_.data_ptr(), bg_.data_ptr(), 8
    d.bs_relu = relu
    if relu == 2:
        d.bs_mask = maskg.data_ptr()
    if two:
        d.bs_z2, d.bs_mean2, d.bs_rstd2, d.bs_z2_cstride = z2g.data_ptr(), m2g.data_ptr(), r2g.data_ptr(), 8
    if split:
        d.in_split, d.in_cstride, d.in2_cstride = 8, 8, 8
    nb = lib.ursn_conv_bs_blocks(ctypes.byref(d))
    assert nb > 0
    for acc in (0, 1):
        partial = torch.full((nb, 3, 8), float("nan"), dtype=torch.float64, device="cuda")
        d.bs_partial = partial.data_ptr()
        base = _rand(rng, x_shape) if acc else np.zeros(x_shape)
        if split:
            dxa, dxb = dev(base[..., :8]), dev(base[..., 8:])
            d.dx2 = dxb.data_ptr()
            _lib.check(lib.ursn_conv_backward_data(ctypes.byref(d), P(dyg), P(wg), P(dxa), acc, stream()))
            torch.cuda.synchronize()
            got = np.concatenate([dxa.cpu().numpy(), dxb.cpu().numpy()], axis=-1)
        else:
            got = conv_backward_data(d, dyg, wg, x_shape, accumulate=acc, dx_init=dev(base)).cpu().numpy()
        want = dx + base
        assert rel_err(got, want) < TOL
        g = want[..., :8] * keep
        sums = partial.cpu().numpy().sum(axis=0)
        ref = np.stack([g.sum(axis=ax), (g * (z - mean) * rstd).sum(axis=ax), (g * (z2 - mean2) * rstd2).sum(axis=ax)])
        scale = np.abs(g).sum(axis=ax).max()
        assert np.abs(sums[:2] - ref[:2]).max() < 1e-6 * scale
        if two:
            assert np.abs(sums[2] - ref[2]).max() < 1e-6 * scale
    # no kernel with this epilogue for 2-D layers or other channel counts: the query says so and the call is refused
    d2 = desc(2, N, S[1:], 8, 8, 3, 1)
    d2.bs_z, d2.bs_mean, d2.bs_rstd, d2.bs_z_cstride, d2.bs_partial = zg.data_ptr(), mg_.data_ptr(), rg_.data_ptr(), 8, partial.data_ptr()
    assert lib.ursn_conv_bs_blocks(ctypes.byref(d2)) == 0
    dxx = torch.empty((N,) + S[1:] + (8,), dtype=torch.float32, device="cuda")
    dy2 = torch.zeros((N,) + S[1:] + (8,), dtype=torch.float32, device="cuda")
    w2 = torch.zeros((3, 3, 8, 8), dtype=torch.float32, device="cuda")
    assert lib.ursn_conv_backward_data(ctypes.byref(d2), P(dy2), P(w2), P(dxx), 0, stream()) != 0


@pytest.mark.parametrize("relu,fused", [(0, 0), (1, 0), (0, 2), (1, 1)])
@pytest.mark.parametrize("S", [(8, 16, 32), (9, 11, 37), (21, 8, 40)])
def test_data_gradient_with_bn_backward_apply_on_load(S, relu, fused):
    """slim.batch_norm backward of an 8 -> 8 layer (lib/resnet_module.py:49, lib/uresnet.py:109) applied WHILE the layer's data
    gradient stages its operand (ursn_conv_desc.vdz_*, ABI 7): the kernel is handed g (gradient at the BatchNorm's output), z
    and the per-channel coefficients, forms dz = A g' + B (z - mu) + C, uses it for dx and stores it for the weight gradient.
    Checked against the oracle's bn_bwd + conv_bwd: dx and the stored dz (every voxel: interior of every tile, each plane
    once), with and without the ReLU mask bn(z) > 0, accumulating, and together with the fused reductions of the NEXT
    BatchNorm backward (fused 1: one z, mask bits; 2: two z)."""
    N = 2
    rng = np.random.default_rng(S[0] * 5 + relu + 3 * fused)
    shp = (N,) + S + (8,)
    w = _rand(rng, (3, 3, 3, 8, 8)) * 0.2
    z = _rand(rng, shp) * 1.7 + 0.8
    g = _rand(rng, shp)
    beta = _rand(rng, (8,)) * 0.4
    ax = (0, 1, 2, 3)
    mean, var = z.mean(axis=ax), z.var(axis=ax)
    rstd = 1.0 / np.sqrt(var + 1e-3)
    f32 = lambda a: np.asarray(a, dtype=np.float32)   # noqa: E731
    S32, T32 = f32(rstd), f32(beta) - f32(mean) * f32(rstd)
    keep = (f32(z) * S32 + T32) > 0 if relu else np.ones(shp, dtype=bool)   # the kernel's (= the forward's) expression
    gm = g * keep
    xhat = (z - mean) * rstd
    c1, c2 = gm.mean(axis=ax), (gm * xhat).mean(axis=ax)
    dz = rstd * (gm - c1 - xhat * c2)                                       # oracle.bn_bwd
    dx = O.conv_bwd(np.zeros(shp), w, 1, dz)[0]
    coef = np.stack([f32(rstd), -(f32(rstd) * f32(rstd)) * f32(c2), -f32(rstd) * f32(c1), f32(mean), S32, T32]).astype(np.float32)
    wg, zg, gg, cg = dev(w), dev(z), dev(g), torch.from_numpy(coef).cuda()
    lib = _lib.load()
    d = desc(3, N, S, 8, 8, 3, 1)
    d.vdz_z, d.vdz_coef, d.vdz_relu = zg.data_ptr(), cg.data_ptr(), relu
    # the consumer of dx: the BatchNorm backward of the layer(s) that produced this layer's input
    z_t, z2_t = _rand(rng, shp) * 1.2 - 0.1, _rand(rng, shp) * 0.6 + 0.2
    mt, rt = z_t.mean(axis=ax), 1.0 / np.sqrt(z_t.var(axis=ax) + 1e-3)
    m2, r2 = z2_t.mean(axis=ax), 1.0 / np.sqrt(z2_t.var(axis=ax) + 1e-3)
    keep_t = rng.random(shp) < 0.55
    keepers = [zt_g := dev(z_t), z2g := dev(z2_t), mtg := dev(mt), rtg := dev(rt), m2g := dev(m2), r2g := dev(r2),
               maskg := torch.from_numpy(_join_mask_words(keep_t).view(np.int64)).cuda()]
    partial = None
    if fused:
        d.bs_z, d.bs_mean, d.bs_rstd, d.bs_z_cstride = zt_g.data_ptr(), mtg.data_ptr(), rtg.data_ptr(), 8
        d.bs_relu, d.bs_mask = 2, maskg.data_ptr()
        if fused == 2:
            d.bs_z2, d.bs_mean2, d.bs_rstd2, d.bs_z2_cstride = z2g.data_ptr(), m2g.data_ptr(), r2g.data_ptr(), 8
        nb = lib.ursn_conv_bs_blocks(ctypes.byref(d))
        assert nb > 0
    buf = ctypes.create_string_buffer(32)
    _lib.check(lib.ursn_conv_plan(ctypes.byref(d), 1, buf, 32))
    assert buf.value == b"tconv"
    for acc in (0, 1):
        dzg = torch.full(shp, float("nan"), dtype=torch.float32, device="cuda")
        d.vdz_out = dzg.data_ptr()
        if fused:
            partial = torch.full((nb, 3, 8), float("nan"), dtype=torch.float64, device="cuda")
            d.bs_partial = partial.data_ptr()
        base = _rand(rng, shp) if acc else np.zeros(shp)
        got = conv_backward_data(d, gg, wg, shp, accumulate=acc, dx_init=dev(base)).cpu().numpy()
        assert lib.ursn_last_kernel_name().startswith(b"tconv_dgrad<8,8>+dz")
        assert rel_err(got, dx + base) < TOL
        assert rel_err(dzg.cpu().numpy(), dz) < 1e-5                      # every voxel written, none twice with another value
        if fused:
            gt = (dx + base) * keep_t
            sums = partial.cpu().numpy().sum(axis=0)
            ref = np.stack([gt.sum(axis=ax), (gt * (z_t - mt) * rt).sum(axis=ax), (gt * (z2_t - m2) * r2).sum(axis=ax)])
            scale = np.abs(gt).sum(axis=ax).max()
            k = 3 if fused == 2 else 2
            assert np.abs(sums[:k] - ref[:k]).max() < 1e-6 * scale
    # no such kernel for 2-D layers or other channel counts: the call is refused
    d2 = desc(3, N, S, 16, 16, 3, 1)
    d2.vdz_z, d2.vdz_coef, d2.vdz_out = zg.data_ptr(), cg.data_ptr(), zg.data_ptr()
    dummy = torch.zeros((N,) + S + (16,), dtype=torch.float32, device="cuda")
    w16 = torch.zeros((3, 3, 3, 16, 16), dtype=torch.float32, device="cuda")
    assert lib.ursn_conv_backward_data(ctypes.byref(d2), P(dummy), P(w16), P(dummy), 0, stream()) != 0
    del keepers


@pytest.mark.parametrize("S", [(10, 7, 70), (8, 12, 128)])
def test_logits_layer_weight_gradient_on_the_vector_pipe(S):
    """conv2 (lib/uresnet.py:94-100, 8 -> 3 channels): its weight gradient runs on v_pk_fma_f32 with a lane per voxel
    (wgrad_valu.hip) when rows are >= 48 voxels wide; dz lives in the net's 4-padded logits buffers."""
    N, ci, co = 2, 8, 3
    rng = np.random.default_rng(S[2])
    x = _rand(rng, (N,) + S + (ci,))
    dy = _rand(rng, (N,) + S + (co,))
    w = np.zeros((3, 3, 3, ci, co))
    _, dw = O.conv_bwd(x, w, 1, dy)
    dyp = np.zeros((N,) + S + (4,))
    dyp[..., :co] = dy
    dyp[..., 3] = 7.0   # the padding channel must not leak into the gradient
    d = desc(3, N, S, ci, co, 3, 1, out_cs=4)
    xg, dyg = dev(x), dev(dyp)
    dwg = conv_backward_weight(d, xg, dyg, w.shape)
    assert rel_err(dwg.cpu().numpy(), dw) < 5e-5
    dwa = conv_backward_weight(d, xg, dyg, w.shape, dw_init=dwg)   # assign_add semantics
    assert rel_err(dwa.cpu().numpy(), 2 * dw) < 5e-5
