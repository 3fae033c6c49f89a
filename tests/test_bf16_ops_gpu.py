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
    ("k3s1_64_64_x8", 3, 2, (8, 8, 8), 64, 64, 3, 1, 0),        # 8-wide level: MFMA tile columns fold over two rows (8-wide boxes)
    ("k3s1_32_64_x16", 3, 1, (10, 12, 16), 32, 64, 3, 1, 0),    # 16-wide level: 16-wide boxes, ragged in z / y
    # z-marching channel-block kernel (bf16_convcb.hip): 16 / 32 contraction channels, ragged 8 x 32 tiles, several z
    # segments, two blocks of produced channels, a produced-channel count that is not a multiple of 32
    ("k3s1_32_32_cb", 3, 2, (9, 21, 37), 32, 32, 3, 1, 0),
    ("k3s1_32_16_cb", 3, 1, (12, 18, 40), 32, 16, 3, 1, 0),     # its data gradient contracts 16 channels
    ("k3s1_32_64_cb", 3, 1, (5, 8, 32), 32, 64, 3, 1, 0),
    ("k3s1_16_24_cb", 3, 2, (7, 11, 33), 16, 24, 3, 1, 0),
    # weight-streaming split-K kernel of the deep levels (bf16_convdeep.hip): >= 64 contraction channels; 8^3 / 16^3 / 6^3
    # levels (split over workgroups where voxels are few), 2C -> C decoder layers (two rounds of chunks), ragged boxes
    ("k3s1_64_64_dp", 3, 4, (8, 8, 8), 64, 64, 3, 1, 0),
    ("k3s1_128_128_dp", 3, 2, (16, 16, 16), 128, 128, 3, 1, 0),
    ("k3s1_256_256_dp", 3, 4, (8, 8, 8), 256, 256, 3, 1, 0),
    ("k3s1_256_256_6_dp", 3, 2, (6, 6, 6), 256, 256, 3, 1, 0),
    ("k3s1_256_128_dp", 3, 1, (8, 8, 8), 256, 128, 3, 1, 0),
    ("k3s1_128_64_rag_dp", 3, 1, (5, 12, 21), 128, 64, 3, 1, 0),
    ("k3s1_64_96_dp", 3, 1, (6, 10, 34), 64, 96, 3, 1, 0),       # produced channels not a multiple of 64; its data gradient contracts 96
    ("k3s2_8_16", 3, 2, (8, 12, 36), 8, 16, 3, 2, 0),
    ("k3s2_odd", 3, 1, (7, 9, 21), 16, 32, 3, 2, 0),            # odd sizes: TF SAME pad-before = 1
    ("k1s1_16_8", 3, 2, (7, 9, 37), 16, 8, 1, 1, 0),            # 1x1 between 8 / 16 channels: operands straight from global memory
    ("k1s1_32_16", 3, 2, (6, 6, 20), 32, 16, 1, 1, 0),
    ("k1s2_8_16", 3, 2, (8, 12, 36), 8, 16, 1, 2, 0),
    ("deconv_16_8", 3, 2, (4, 6, 18), 16, 8, 3, 2, 1),
    ("deconv_64_32", 3, 1, (3, 4, 6), 64, 32, 3, 2, 1),
    # single-launch stride-2 scatter passes of the deeper levels (bf16_scatter.hip): transposed convs forward, stride-2 convs'
    # data gradient; every kernel form <produced-channel tiles, voxel tiles>, two blocks of produced channels, ragged boxes
    ("deconv_32_16_sc", 3, 2, (6, 8, 12), 32, 16, 3, 2, 1),           # <1,4>
    ("deconv_32_16_big_sc", 3, 2, (32, 32, 32), 32, 16, 3, 2, 1),     # <1,8>: 128-voxel boxes
    ("deconv_64_32_big_sc", 3, 2, (32, 32, 32), 64, 32, 3, 2, 1),     # <2,8>
    ("deconv_128_64_sc", 3, 1, (5, 8, 8), 128, 64, 3, 2, 1),          # <4,4>
    ("deconv_256_128_sc", 3, 2, (4, 4, 4), 256, 128, 3, 2, 1),        # <4,4>, two blocks of produced channels, 64 KB image
    ("k3s2_16_32_sc", 3, 2, (8, 12, 20), 16, 32, 3, 2, 0),
    ("k3s2_32_64_sc", 3, 1, (10, 12, 24), 32, 64, 3, 2, 0),
    ("k3s2_128_256_sc", 3, 1, (8, 8, 8), 128, 256, 3, 2, 0),
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
    if tag.endswith("_cb"):   # the dispatch really took the channel-block kernel
        lib.ursn_last_kernel_name.restype = ctypes.c_char_p
        assert lib.ursn_last_kernel_name().startswith(b"bcbconv_bf16"), lib.ursn_last_kernel_name()
    if tag.endswith("_dp"):   # ... the deep-level kernel
        lib.ursn_last_kernel_name.restype = ctypes.c_char_p
        assert lib.ursn_last_kernel_name().startswith(b"bdconv_bf16"), lib.ursn_last_kernel_name()
    if tag.endswith("_sc") and tr:   # ... the single-launch scatter kernel
        lib.ursn_last_kernel_name.restype = ctypes.c_char_p
        assert lib.ursn_last_kernel_name().startswith(b"bsconv_bf16"), lib.ursn_last_kernel_name()
    dxg = torch.full(x.shape, float("nan"), dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.ursn_conv_backward_data(ctypes.byref(d), P(dyg), P(wg), P(dxg), 0, stream()))
    torch.cuda.synchronize()
    e = np.abs(dxg.float().cpu().numpy() - dx).max() / np.abs(dx).max()
    assert e <= BF_TOL, ("dgrad", e)
    if tag.endswith("_sc") and not tr:
        lib.ursn_last_kernel_name.restype = ctypes.c_char_p
        assert lib.ursn_last_kernel_name().startswith(b"bsconv_bf16"), lib.ursn_last_kernel_name()
    if tag.endswith("_dp") and co % 64 == 0:   # the contraction splits over 2 | 4 waves: an even number of 32-channel chunks
        assert lib.ursn_last_kernel_name().startswith(b"bdconv_bf16"), lib.ursn_last_kernel_name()
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
    if tag.endswith("_dp") and co % 64 == 0:   # the deep-level weight-gradient kernel (bf16_wgraddeep.hip)
        assert lib.ursn_last_kernel_name().startswith(b"bdwgrad_bf16"), lib.ursn_last_kernel_name()


@pytest.mark.parametrize("case", [c for c in CASES if not c[8]][:29], ids=[c[0] for c in CASES if not c[8]][:29])
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


# ---- fused forms of the bf16 plan at op level: the variants cfg5's default plan runs, each against the fp64
# oracle on the operands the kernel reads, one bf16 ulp (2^-8 of max) for bf16 outputs / 2e-5 of max for fp32 filter gradients.
from _insitu import staged_bn


def _f32(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def _bf_w(w):
    """fp32 master weights whose values are bf16-representable (the pack kernels round them; the oracle reads the same)."""
    w32 = torch.from_numpy(np.asarray(w, np.float32)).to(torch.bfloat16).float().numpy()
    return w32.astype(np.float64), _f32(w32)


def _wgrad(lib, d, x_t, dy_t, wshape):
    nb = lib.ursn_conv_wgrad_scratch_bytes(ctypes.byref(d))
    scratch = torch.empty(nb + 256, dtype=torch.uint8, device="cuda")
    dw = torch.zeros(wshape, dtype=torch.float32, device="cuda")
    _lib.check(lib.ursn_conv_backward_weight(ctypes.byref(d), P(x_t), P(dy_t), P(dw), P(scratch), nb, stream()))
    torch.cuda.synchronize()
    return dw.cpu().numpy()


AFF_CASES = [
    # tag, N, S, C, relu
    ("8_8_plain_odd", 2, (9, 21, 37), 8, 0),          # resnet_conv1 -> resnet_conv2 (lib/resnet_module.py:43-66): BatchNorm, no activation
    ("8_8_relu_zseg", 1, (40, 18, 34), 8, 1),         # conv1 -> conv2 (lib/uresnet.py:103-121): BatchNorm + ReLU; several z segments
    ("16_16_plain_zseg", 1, (36, 9, 40), 16, 0),
    ("16_16_relu_odd", 2, (11, 19, 35), 16, 1),
]


@pytest.mark.parametrize("case", AFF_CASES, ids=[c[0] for c in AFF_CASES])
def test_bf16_normalise_on_load_forward_statistics_and_weight_gradient(case):
    tag, N, S, C, relu = case
    lib = _lib.load()
    rng = np.random.default_rng(101 + C + relu)
    z, zg = bf(rng.standard_normal((N,) + S + (C,)) * 1.7 + 0.8)          # the raw output of the producing conv
    mean = (rng.standard_normal(C) * 0.5 + 0.8).astype(np.float32)
    rstd = rng.uniform(0.4, 1.6, C).astype(np.float32)
    beta = (rng.standard_normal(C) * 0.3).astype(np.float32)
    x = staged_bn(z, mean, rstd, beta, relu)                               # what the kernel stages: bf16(fma(z, r, fma(-mu, r, beta)))
    w, wg = _bf_w(rng.standard_normal((3, 3, 3, C, C)) * 0.2)
    y = O.conv_fwd(x, w, 1)
    d = desc(3, N, S, C, C, 3, 1)
    d.dtype = 1
    mg, rg, bg = _f32(mean), _f32(rstd), _f32(beta)
    d.in_mean, d.in_rstd, d.in_beta, d.in_relu = mg.data_ptr(), rg.data_ptr(), bg.data_ptr(), relu
    yg = torch.full(y.shape, float("nan"), dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.ursn_conv_forward(ctypes.byref(d), P(zg), P(wg), P(yg), stream()))
    torch.cuda.synchronize()
    e = np.abs(yg.float().cpu().numpy() - y).max() / np.abs(y).max()
    assert e <= BF_TOL, ("forward", e)
    # ... with the fused moments of the STORED output
    y2 = torch.empty_like(yg)
    om, orr = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    nb = 1 << 24
    scratch = torch.empty(nb, dtype=torch.uint8, device="cuda")
    _lib.check(lib.ursn_conv_forward_stats(ctypes.byref(d), P(zg), P(wg), P(y2), P(om), P(orr), 1e-3, P(scratch), nb, stream()))
    torch.cuda.synchronize()
    assert torch.equal(y2, yg)
    ys = y2.float().cpu().numpy().astype(np.float64)
    ax = tuple(range(ys.ndim - 1))
    assert np.abs(om.cpu().numpy() - ys.mean(axis=ax)).max() < 1e-5 * np.sqrt(ys.var(axis=ax).max()) + 1e-6
    assert rel_err(orr.cpu().numpy(), 1 / np.sqrt(ys.var(axis=ax) + 1e-3)) < 1e-5
    # weight gradient reads the same staged activation
    dy, dyg = bf(rng.standard_normal(y.shape))
    dw = O.conv_bwd(x, w, 1, dy)[1]
    assert rel_err(_wgrad(lib, d, zg, dyg, (3, 3, 3, C, C)), dw) < 2e-5


@pytest.mark.parametrize("split", [0, 1], ids=["one_tensor", "two_tensors"])
@pytest.mark.parametrize("shape", [(2, (9, 21, 37)), (1, (40, 10, 34))], ids=["odd", "zseg"])
def test_bf16_data_gradient_with_fused_shortcut_term_and_split_output(shape, split):
    """First decoder unit of level 0 (lib/resnet_module.py:25-51, 16 -> 8): d(in) = conv1^T(dz1) + shortcut^T(dz_sc) in one
    kernel; with F = 8 the plan writes the two halves of the concat gradient as two tensors."""
    N, S = shape
    lib = _lib.load()
    rng = np.random.default_rng(7 + split + N)
    w, wg = _bf_w(rng.standard_normal((3, 3, 3, 16, 8)) * 0.2)
    pw, pwg = _bf_w(rng.standard_normal((16, 8)) * 0.3)
    dy, dyg = bf(rng.standard_normal((N,) + S + (8,)))
    pdy, pdyg = bf(rng.standard_normal((N,) + S + (8,)))
    dummy = np.zeros((N,) + S + (16,))
    dx = O.conv_bwd(dummy, w, 1, dy)[0] + pdy @ pw.T
    d = desc(3, N, S, 16, 8, 3, 1)
    d.dtype = 1
    d.pw_dy, d.pw_w = pdyg.data_ptr(), pwg.data_ptr()
    if split:
        d.in_split = 8
        a = torch.full((N,) + S + (8,), float("nan"), dtype=torch.bfloat16, device="cuda")
        b = torch.full((N,) + S + (8,), float("nan"), dtype=torch.bfloat16, device="cuda")
        d.dx2 = b.data_ptr()
        _lib.check(lib.ursn_conv_backward_data(ctypes.byref(d), P(dyg), P(wg), P(a), 0, stream()))
        torch.cuda.synchronize()
        got = np.concatenate([a.float().cpu().numpy(), b.float().cpu().numpy()], axis=-1)
        assert np.abs(got - dx).max() <= BF_TOL * np.abs(dx).max()
        return
    g = torch.full(dx.shape, float("nan"), dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.ursn_conv_backward_data(ctypes.byref(d), P(dyg), P(wg), P(g), 0, stream()))
    torch.cuda.synchronize()
    assert np.abs(g.float().cpu().numpy() - dx).max() <= BF_TOL * np.abs(dx).max()
    base, baseg = bf(rng.standard_normal(dx.shape))
    _lib.check(lib.ursn_conv_backward_data(ctypes.byref(d), P(dyg), P(wg), P(baseg), 1, stream()))
    torch.cuda.synchronize()
    assert np.abs(baseg.float().cpu().numpy() - (dx + base)).max() <= BF_TOL * np.abs(dx + base).max()


@pytest.mark.parametrize("shape", [(2, (9, 21, 37)), (1, (40, 18, 34)), (2, (8, 16, 64)), (1, (5, 9, 130))], ids=["odd", "zseg", "even", "three_tiles"])
def test_bf16_conv0_reads_the_fp32_scalar_input(shape):
    """conv0 (lib/uresnet.py:37-45) on the raw fp32 data, one channel per voxel: forward (+ fused moments) and weight gradient."""
    N, S = shape
    lib = _lib.load()
    rng = np.random.default_rng(13 + N)
    x32 = (rng.uniform(0, 4, (N,) + S + (1,)) * (rng.uniform(0, 1, (N,) + S + (1,)) > 0.6)).astype(np.float32)
    x = O.bf16_round(x32.astype(np.float64))            # staged as (bf16(value), 0 x 7)
    w, wg = _bf_w(rng.standard_normal((3, 3, 3, 1, 8)) * 0.3)
    y = O.conv_fwd(x, w, 1)
    d = desc(3, N, S, 1, 8, 3, 1)
    d.dtype = 1
    xg = _f32(x32)
    yg = torch.full(y.shape, float("nan"), dtype=torch.bfloat16, device="cuda")
    om, orr = torch.empty(8, device="cuda"), torch.empty(8, device="cuda")
    nb = 1 << 24
    scratch = torch.empty(nb, dtype=torch.uint8, device="cuda")
    _lib.check(lib.ursn_conv_forward_stats(ctypes.byref(d), P(xg), P(wg), P(yg), P(om), P(orr), 1e-3, P(scratch), nb, stream()))
    torch.cuda.synchronize()
    assert np.abs(yg.float().cpu().numpy() - y).max() <= BF_TOL * np.abs(y).max()
    ys = yg.float().cpu().numpy().astype(np.float64)
    ax = tuple(range(ys.ndim - 1))
    assert rel_err(orr.cpu().numpy(), 1 / np.sqrt(ys.var(axis=ax) + 1e-3)) < 1e-5
    assert rel_err(om.cpu().numpy(), ys.mean(axis=ax)) < 1e-5
    lib.ursn_last_kernel_name.restype = ctypes.c_char_p
    assert lib.ursn_last_kernel_name().startswith(b"b0conv_bf16"), lib.ursn_last_kernel_name()   # the taps-as-contraction kernel
    dy, dyg = bf(rng.standard_normal(y.shape))
    dw = O.conv_bwd(x, w, 1, dy)[1]
    assert rel_err(_wgrad(lib, d, xg, dyg, (3, 3, 3, 1, 8)), dw) < 2e-5
    assert lib.ursn_last_kernel_name().startswith(b"b0wgrad_bf16"), lib.ursn_last_kernel_name()


@pytest.mark.parametrize("shape", [(2, (8, 12, 36)), (1, (16, 20, 68))], ids=["small", "ragged_tiles"])
def test_bf16_stride2_scatter_passes_on_channel_strided_views(shape):
    """The single-launch stride-2 scatter kernel (bf16_deconv3.hip) writing an 8-channel slice of a 16-channel concat voxel:
    the transposed conv's forward output (lib/uresnet.py:66-88) and the stride-2 conv's data gradient ACCUMULATED into the
    skip half of a concat gradient (lib/resnet_module.py:43-51 with stride 2)."""
    N, S = shape
    lib = _lib.load()
    rng = np.random.default_rng(23 + N)
    # transposed conv 16 -> 8, coarse S -> fine 2S, output = channels [0, 8) of a 16-channel buffer
    x, xg = bf(rng.standard_normal((N,) + S + (16,)))
    w, wg = _bf_w(rng.standard_normal((3, 3, 3, 8, 16)) * 0.2)
    y = O.deconv_fwd(x, w)
    d = desc(3, N, S, 16, 8, 3, 2, transposed=1, out_cs=16)
    d.dtype = 1
    keep, buf = bf(rng.standard_normal(y.shape[:-1] + (16,)))
    _lib.check(lib.ursn_conv_forward(ctypes.byref(d), P(xg), P(wg), P(buf), stream()))
    torch.cuda.synchronize()
    got = buf.float().cpu().numpy()
    assert np.abs(got[..., :8] - y).max() <= BF_TOL * np.abs(y).max()
    assert np.array_equal(got[..., 8:], keep[..., 8:])                 # the other half of the voxel is untouched
    # stride-2 conv 8 -> 16 on the fine grid 2S: its data gradient accumulated into channels [8, 16) of a 16-channel gradient
    fine = tuple(2 * s for s in S)
    w2, w2g = _bf_w(rng.standard_normal((3, 3, 3, 8, 16)) * 0.2)
    dy, dyg = bf(rng.standard_normal((N,) + S + (16,)))
    dx = O.conv_bwd(np.zeros((N,) + fine + (8,)), w2, 2, dy)[0]
    base, baseg = bf(rng.standard_normal((N,) + fine + (16,)))
    d2 = desc(3, N, fine, 8, 16, 3, 2, in_cs=16)
    d2.dtype = 1
    view = baseg.view(-1)[8:]                                             # channel offset 8 inside the 16-channel voxel
    _lib.check(lib.ursn_conv_backward_data(ctypes.byref(d2), P(dyg), P(w2g), ctypes.c_void_p(view.data_ptr()), 1, stream()))
    torch.cuda.synchronize()
    got = baseg.float().cpu().numpy()
    want = base[..., 8:] + dx
    assert np.abs(got[..., 8:] - want).max() <= BF_TOL * np.abs(want).max()
    assert np.array_equal(got[..., :8], base[..., :8])


def _bn_desc(V, C, relu):
    d = _lib.ursn_bn_bf16_desc()
    d.voxels, d.channels, d.relu = V, C, relu
    return d


def _mask_bytes(y):
    """one byte per 16-byte piece: bit j = (channel j of the piece > 0)"""
    m = (y.reshape(-1, y.shape[-1] // 8, 8) > 0).astype(np.uint8)
    return (m << np.arange(8, dtype=np.uint8)).sum(axis=-1).astype(np.uint8)


@pytest.mark.parametrize("C", [8, 16, 32])
@pytest.mark.parametrize("form", ["join_conv_shortcut", "join_identity"])
def test_bf16_join_forward_mask_bytes_and_backward(C, form):
    """The residual join (lib/resnet_module.py:68) in the bf16 plan: out = relu(bn(z) + bn2(z2) | + x) with the ReLU mask
    BYTES written beside it, and its backward reading the mask bytes instead of out: dz (and dz2 / the identity share dres,
    overwritten and accumulated), d(beta) (+ d(beta2))."""
    lib = _lib.load()
    V = 3 * 7 * 9 * 41
    rng = np.random.default_rng(C + len(form))
    z, zg = bf(rng.standard_normal((V, C)) * 1.5 + 0.3)
    mu, var = z.mean(0), z.var(0)
    mean, rstd = mu.astype(np.float32), (1 / np.sqrt(var + 1e-3)).astype(np.float32)
    beta = (rng.standard_normal(C) * 0.3).astype(np.float32)
    bn = lambda zz, m, r, b: (zz - m.astype(np.float64)) * r.astype(np.float64) + b.astype(np.float64)
    d = _bn_desc(V, C, 1)
    keep = [_f32(mean), _f32(rstd), _f32(beta)]
    d.z, d.mean, d.rstd, d.beta = zg.data_ptr(), keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr()
    if form == "join_conv_shortcut":
        z2, z2g = bf(rng.standard_normal((V, C)) * 0.7 - 0.2)
        mean2, rstd2 = z2.mean(0).astype(np.float32), (1 / np.sqrt(z2.var(0) + 1e-3)).astype(np.float32)
        beta2 = (rng.standard_normal(C) * 0.3).astype(np.float32)
        keep += [_f32(mean2), _f32(rstd2), _f32(beta2)]
        d.z2, d.mean2, d.rstd2, d.beta2 = z2g.data_ptr(), keep[3].data_ptr(), keep[4].data_ptr(), keep[5].data_ptr()
        pre = bn(z, mean, rstd, beta) + bn(z2, mean2, rstd2, beta2)
    else:
        res, resg = bf(rng.standard_normal((V, C)))
        d.res = resg.data_ptr()
        pre = bn(z, mean, rstd, beta) + res
    y = np.maximum(pre, 0.0)
    yg = torch.full((V, C), float("nan"), dtype=torch.bfloat16, device="cuda")
    mk = torch.zeros((V, C // 8), dtype=torch.uint8, device="cuda")
    d.y, d.mask_out = yg.data_ptr(), mk.data_ptr()
    _lib.check(lib.ursn_bn_bf16_forward(ctypes.byref(d), stream()))
    torch.cuda.synchronize()
    ys = yg.float().cpu().numpy().astype(np.float64)
    assert np.abs(ys - y).max() <= BF_TOL * np.abs(y).max()
    assert np.array_equal(mk.cpu().numpy(), _mask_bytes(ys))            # the mask describes the STORED output
    # backward from the mask bytes
    dy, dyg = bf(rng.standard_normal((V, C)))
    g = dy * (ys > 0)
    xh = (z - mean.astype(np.float64)) * rstd.astype(np.float64)
    dz = rstd.astype(np.float64) * (g - g.mean(0) - xh * (g * xh).mean(0))
    b = _bn_desc(V, C, 1)
    b.z, b.mean, b.rstd, b.beta = d.z, d.mean, d.rstd, d.beta
    b.dy, b.mask = dyg.data_ptr(), mk.data_ptr()
    dzg = torch.full((V, C), float("nan"), dtype=torch.bfloat16, device="cuda")
    dbeta = torch.zeros(C, device="cuda")
    b.dz, b.dbeta = dzg.data_ptr(), dbeta.data_ptr()
    nb = lib.ursn_bn_bf16_scratch_bytes(V, C)
    scratch = torch.empty(nb, dtype=torch.uint8, device="cuda")
    if form == "join_conv_shortcut":
        xh2 = (z2 - mean2.astype(np.float64)) * rstd2.astype(np.float64)
        dz2 = rstd2.astype(np.float64) * (g - g.mean(0) - xh2 * (g * xh2).mean(0))
        dz2g = torch.full((V, C), float("nan"), dtype=torch.bfloat16, device="cuda")
        dbeta2 = torch.zeros(C, device="cuda")
        b.z2, b.mean2, b.rstd2 = d.z2, d.mean2, d.rstd2
        b.dz2, b.dbeta2 = dz2g.data_ptr(), dbeta2.data_ptr()
        _lib.check(lib.ursn_bn_bf16_backward(ctypes.byref(b), P(scratch), nb, stream()))
        torch.cuda.synchronize()
        assert np.abs(dz2g.float().cpu().numpy() - dz2).max() <= BF_TOL * np.abs(dz2).max()
        assert np.abs(dbeta2.cpu().numpy() - g.sum(0)).max() <= 1e-5 * np.abs(g).sum(0).max()
    else:
        for acc in (0, 1):
            base, baseg = bf(rng.standard_normal((V, C)))
            b.dres, b.dres_accumulate = baseg.data_ptr(), acc
            dbeta.zero_()
            _lib.check(lib.ursn_bn_bf16_backward(ctypes.byref(b), P(scratch), nb, stream()))
            torch.cuda.synchronize()
            want = g + (base if acc else 0.0)
            assert np.abs(baseg.float().cpu().numpy() - want).max() <= BF_TOL * np.abs(want).max(), acc
    assert np.abs(dzg.float().cpu().numpy() - dz).max() <= BF_TOL * np.abs(dz).max()
    assert np.abs(dbeta.cpu().numpy() - g.sum(0)).max() <= 1e-5 * np.abs(g).sum(0).max()


def test_bf16_concat_pass_and_two_operand_backward_of_conv0():
    """F = 8 (cfg5): the level-0 concat voxel [relu(bn(z_deconv)) | relu(bn(z_conv0))] written whole in one pass, and conv0's
    BatchNorm backward summing its two gradient contributions (encoder share + skip share) with the mask bn(z) > 0."""
    lib = _lib.load()
    V, C = 2 * 9 * 11 * 37, 8
    rng = np.random.default_rng(5)
    z, zg = bf(rng.standard_normal((V, C)) * 1.3 + 0.2)
    z2, z2g = bf(rng.standard_normal((V, C)) * 0.9 - 0.1)
    st = []
    for t in (z, z2):
        st += [t.mean(0).astype(np.float32), (1 / np.sqrt(t.var(0) + 1e-3)).astype(np.float32), (rng.standard_normal(C) * 0.3).astype(np.float32)]
    dev = [_f32(a) for a in st]
    bn = lambda zz, m, r, b: (zz - m.astype(np.float64)) * r.astype(np.float64) + b.astype(np.float64)
    d = _bn_desc(V, C, 1)
    d.z, d.mean, d.rstd, d.beta = zg.data_ptr(), dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr()
    d.z2, d.mean2, d.rstd2, d.beta2 = z2g.data_ptr(), dev[3].data_ptr(), dev[4].data_ptr(), dev[5].data_ptr()
    d.cat = 1
    yg = torch.full((V, 16), float("nan"), dtype=torch.bfloat16, device="cuda")
    d.y, d.y_cstride = yg.data_ptr(), 16
    _lib.check(lib.ursn_bn_bf16_forward(ctypes.byref(d), stream()))
    torch.cuda.synchronize()
    want = np.concatenate([np.maximum(bn(z, *st[0:3]), 0), np.maximum(bn(z2, *st[3:6]), 0)], axis=-1)
    assert np.abs(yg.float().cpu().numpy() - want).max() <= BF_TOL * np.abs(want).max()
    # conv0's backward: g = (dy + dy2) * (bn(z) > 0)
    dy, dyg = bf(rng.standard_normal((V, C)))
    dy2, dy2g = bf(rng.standard_normal((V, C)))
    g = (dy + dy2) * (bn(z, *st[0:3]) > 0)
    xh = (z - st[0].astype(np.float64)) * st[1].astype(np.float64)
    dz = st[1].astype(np.float64) * (g - g.mean(0) - xh * (g * xh).mean(0))
    b = _bn_desc(V, C, 1)
    b.z, b.mean, b.rstd, b.beta = d.z, d.mean, d.rstd, d.beta
    b.dy, b.dy2 = dyg.data_ptr(), dy2g.data_ptr()
    dzg = torch.full((V, C), float("nan"), dtype=torch.bfloat16, device="cuda")
    dbeta = torch.zeros(C, device="cuda")
    b.dz, b.dbeta = dzg.data_ptr(), dbeta.data_ptr()
    nb = lib.ursn_bn_bf16_scratch_bytes(V, C)
    scratch = torch.empty(nb, dtype=torch.uint8, device="cuda")
    _lib.check(lib.ursn_bn_bf16_backward(ctypes.byref(b), P(scratch), nb, stream()))
    torch.cuda.synchronize()
    assert np.abs(dzg.float().cpu().numpy() - dz).max() <= BF_TOL * np.abs(dz).max()
    assert np.abs(dbeta.cpu().numpy() - g.sum(0)).max() <= 1e-5 * np.abs(g).sum(0).max()


@pytest.mark.parametrize("chan", [(32, 16), (64, 32)], ids=["32_16", "64_32"])
@pytest.mark.parametrize("shape", [(2, (9, 21, 37)), (1, (12, 8, 64))], ids=["odd", "zseg"])
def test_bf16_channel_block_data_gradient_with_fused_shortcut_term(shape, chan):
    """First decoder unit of levels 1 / 2 (lib/resnet_module.py:25-51, 32 -> 16 and 64 -> 32): d(in) = conv1^T(dz1) +
    shortcut^T(dz_sc) in the z-marching channel-block kernel (the shortcut's dz plane staged beside the x plane, its weights as
    KS more A fragments of the centre tap); overwrite and accumulate."""
    N, S = shape
    ci, co = chan
    lib = _lib.load()
    rng = np.random.default_rng(ci + N)
    w, wg = _bf_w(rng.standard_normal((3, 3, 3, ci, co)) * 0.2)
    pw, pwg = _bf_w(rng.standard_normal((ci, co)) * 0.3)
    dy, dyg = bf(rng.standard_normal((N,) + S + (co,)))
    pdy, pdyg = bf(rng.standard_normal((N,) + S + (co,)))
    dx = O.conv_bwd(np.zeros((N,) + S + (ci,)), w, 1, dy)[0] + pdy @ pw.T
    d = desc(3, N, S, ci, co, 3, 1)
    d.dtype = 1
    d.pw_dy, d.pw_w = pdyg.data_ptr(), pwg.data_ptr()
    g = torch.full(dx.shape, float("nan"), dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.ursn_conv_backward_data(ctypes.byref(d), P(dyg), P(wg), P(g), 0, stream()))
    torch.cuda.synchronize()
    lib.ursn_last_kernel_name.restype = ctypes.c_char_p
    assert lib.ursn_last_kernel_name().endswith(b"+pw") and lib.ursn_last_kernel_name().startswith(b"bcbconv"), lib.ursn_last_kernel_name()
    assert np.abs(g.float().cpu().numpy() - dx).max() <= BF_TOL * np.abs(dx).max()
    base, baseg = bf(rng.standard_normal(dx.shape))
    _lib.check(lib.ursn_conv_backward_data(ctypes.byref(d), P(dyg), P(wg), P(baseg), 1, stream()))
    torch.cuda.synchronize()
    assert np.abs(baseg.float().cpu().numpy() - (dx + base)).max() <= BF_TOL * np.abs(dx + base).max()
