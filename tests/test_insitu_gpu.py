"""In-situ parity of a full-depth training step, both plans: after one accum_gradients every layer's outputs
(z, statistics, activation, dz, dbeta, dw, the summed data gradient of each activation) are recomputed with the ORACLE's
primitives from the product's own stored operands (tests/_insitu.py) and held to kernel-level tolerances -- fp32: 2e-5 of
max (1e-5 for the elementwise passes); bf16: one bf16 ulp + 2e-5 of max for bf16-stored tensors, 2e-5 of max for the fp32
filter gradients.  These checks do not depend on ReLU / rounding flips upstream, so they are the tight guard behind the
conditioning-limited net-level bounds of test_configs_gpu.py / test_bf16_net_gpu.py.
Layers: lib/uresnet.py:37-121, lib/resnet_module.py:25-68.  PARITY UNPINNED (oracle/__init__.py)."""
import numpy as np
import pytest

from _insitu import FullSize, InSitu
from _net import as_f32_exact, make_inputs, oracle_params
from uresnet_amd import uresnet

pytestmark = pytest.mark.gpu

CASES = [
    # tag, dims, F, classes, batch, num_strides, precision
    ("cfg3_model_3d64_fp32", (64, 64, 64, 1), 8, 3, 2, 5, "fp32"),
    ("cfg3_model_3d128_fp32", (128, 128, 128, 1), 8, 3, 1, 5, "fp32"),
    ("cfg2_model_2d128_fp32", (128, 128, 1), 16, 5, 2, 5, "fp32"),
    ("cfg5_model_3d64_bf16", (64, 64, 64, 1), 8, 3, 2, 5, "bf16"),
    ("f16_3d32x32x64_ns3_bf16", (32, 32, 64, 1), 16, 5, 1, 3, "bf16"),   # 16-channel level 0: concat buffer plan, generic conv0
    ("2d_f16_ns5_bf16", (64, 64, 1), 16, 3, 2, 5, "bf16"),
]


def _step(dims, base, ncls, N, ns, prec, seed=37):
    P = as_f32_exact(oracle_params(dims, base, ncls, num_strides=ns))
    data, label, weight = make_inputs(dims, ncls, N, seed=seed)
    net = uresnet(dims=list(dims), num_class=ncls, base_num_outputs=base, num_strides=ns)
    net.construct(trainable=True, use_weight=True, learning_rate=1e-3, precision=prec)
    net.set_variables(P)
    net.zero_gradients(None)
    res, _ = net.accum_gradients(None, data, label, weight)
    assert np.isfinite(res[1])
    return net, P, data, label, weight


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_every_layer_against_oracle_primitives_on_stored_operands(case):
    tag, dims, base, ncls, N, ns, prec = case
    net, P, data, label, weight = _step(dims, base, ncls, N, ns, prec)
    chk = InSitu(net, P, dims, base, ncls, ns, data, label, weight, bf16=(prec == "bf16"))
    T = chk.run(tag)
    assert set(T.worst) >= {"z", "rstd", "act", "join", "dlogits", "dz", "dw", "dx"}, sorted(T.worst)


# ---- the optional kernel paths behind the A/B switches, held to the same local bounds ------------------------------------
_CHILD = r"""
import sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from test_insitu_gpu import _step
from _insitu import InSitu
dims, base, ncls, N, ns, prec = (32, 64, 64, 1), 8, 3, 1, 5, sys.argv[2]
net, P, data, label, weight = _step(dims, base, ncls, N, ns, prec)
T = InSitu(net, P, dims, base, ncls, ns, data, label, weight, bf16=(prec == "bf16")).run(sys.argv[3])
assert set(T.worst) >= {"z", "rstd", "act", "join", "dlogits", "dz", "dw", "dx"}
print("INSITU_OK")
"""
SWITCHES = [
    ("bf16", {"URSN_B3CONV_DMA": "0", "URSN_BF16_SKIP0_MERGE": "0", "URSN_BF16_PREPACK": "0", "URSN_BF16_FWD_OVERLAP": "0"}),   # register-staged planes instead of the LDS-DMA ring; the skip's gradient share in its own tensor; per-launch weight packing; shortcut convs in line
    ("bf16", {"URSN_BSCONV": "0", "URSN_B0CONV": "0", "URSN_BS2K8": "0", "URSN_B3CONV_ACC_DMA": "0", "URSN_BDWGRAD": "0", "URSN_BF16_RESIDUAL_IN_DGRAD": "0", "URSN_BF16_HEAD_BN_BWD": "0", "URSN_BBN_CAT_PAIRS": "0"}),   # round-4 kernels off: stride-2 scatter passes by parity class, conv0 as an 8-channel layer
    ("bf16", {"URSN_BCB": "0", "URSN_SLAB_FOLD": "0"}),    # generic box kernel at levels 1-2, one-stage slab reduce
    ("bf16", {"URSN_BF16_FUSE_BN_BWD_REDUCE": "1"}),       # BatchNorm-backward reductions in the data-gradient epilogue
    ("bf16", {"URSN_B3CONV_PW": "0", "URSN_BF16_NORM_ON_LOAD": "0", "URSN_BF16_SKIP0_OWN": "0"}),   # no fused shortcut term, materialised activations, skip inside the concat buffer
    ("bf16", {"URSN_B3CONV": "0", "URSN_B3WGRAD": "0", "URSN_BDECONV": "0", "URSN_BPW": "0"}),      # generic kernels everywhere
    ("fp32", {"URSN_FUSE_BN_BWD_REDUCE": "0", "URSN_FUSE_SHORTCUT_DGRAD": "0", "URSN_FUSE_SHORTCUT_DGRAD_S2": "0", "URSN_RELU_MASK": "0"}),   # (the stride-2 shortcut's data gradient as its own pass too)
    ("fp32", {"URSN_SPLIT_CAT": "0", "URSN_NORM_ON_LOAD": "0"}),    # concat buffer, materialised resnet_conv1 activations
    ("fp32", {"URSN_DISABLE_TILED": "1", "URSN_WGRAD_STREAM": "0"}),
    ("fp32", {"URSN_S2CONV_V2": "2", "URSN_HEAD_BN_BWD": "0", "URSN_WGRADZ_OCC3": "1"}),   # stride-2 gather kernel with 16-byte operand reads on every layer it can take; logits-layer BatchNorm-backward sums as a separate pass
    ("fp32", {"URSN_WGRADQ": "1", "URSN_NORM_ON_LOAD": "2"}),   # 4x4-block weight gradient at level 0 (opt-in, wgradq_tiled_kernel.h)
]


@pytest.mark.parametrize("case", SWITCHES, ids=["%s:%s" % (p, ",".join("%s=%s" % kv for kv in sorted(e.items()))) for p, e in SWITCHES])
def test_optional_kernel_paths_in_situ(case):
    """cfg5's / cfg3's model at 32 x 64 x 64 with the A/B switches of DESIGN.md section 3 flipped (child process: the switches are
    read once per process): every alternative path is held to the same per-layer bounds as the default plan."""
    import os
    import subprocess
    import sys
    prec, env = case
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", _CHILD, root, prec, "switch"], env=dict(os.environ, **env), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0 and "INSITU_OK" in p.stdout, (p.stdout[-1500:], p.stderr[-2500:])


# ---- full size ---------------------------------------------------------------------------------------------------------
U = "UResNet/"


def _full_size_checks(net, P, bf16, ns=5):
    """Level-0 / level-1 layers of a full-size step: 8 -> 8 with a normalised-on-load input (bf16) and the join's gradient,
    conv1 (ReLU mask), the split-input 16 -> 8 pair of the first decoder unit at level 0 with the fused shortcut term, the
    16 -> 16 layer of level 1 behind an identity shortcut, and the last transposed conv."""
    fs = FullSize(net, P, bf16, workers=14)
    m9, m8 = U + "resnet_module%d" % (ns + 4), U + "resnet_module%d" % (ns + 3)
    dec = U + "deconv%d" % (ns - 1)
    q = fs.q

    def stored_or_bn(name, relu):
        """rows of activation `name`; when the plan never writes it, what its consumers stage (FullSize.staged)"""
        try:
            a = fs.t(name)
            return lambda n, lo, hi: a[n, lo:hi].astype(np.float64)
        except Exception as e:
            assert "not materialised" in str(e), e
            return lambda n, lo, hi: fs.staged(name, n, lo, hi, relu)

    # A: module2/resnet_conv2 of the last decoder unit (8 -> 8), x = bn(z of resnet_conv1), g = the join's masked gradient
    c1, c2, sc = m9 + "/module2/resnet_conv1", m9 + "/module2/resnet_conv2", None
    x_a = stored_or_bn(c1, False)
    out, gout = fs.t(m9 + "/module2"), fs.t(m9 + "/module2:grad")
    g_join = lambda n, a, b: gout[n, a:b].astype(np.float64) * (out[n, a:b] > 0)
    fs.check_stats(c2)
    fs.check_forward(c2, "conv", x_a)
    fs.check_bn_backward(c2, g_join)
    fs.check_weight_gradient(c2, "conv", x_a)
    fs.check_data_gradient(c1 + ":grad", fs.t(c1 + ":grad"), [(c2, "conv", slice(None))])
    # ... and resnet_conv1 behind it: no activation, identity shortcut: d(in) = conv1^T(dz1) + g_join
    g1 = fs.t(c1 + ":grad")
    fs.check_bn_backward(c1, lambda n, a, b: g1[n, a:b].astype(np.float64))
    x_in = stored_or_bn(m9 + "/module1", False)
    fs.check_forward(c1, "conv", x_in)
    fs.check_weight_gradient(c1, "conv", x_in)
    fs.check_data_gradient(m9 + "/module1:grad", fs.t(m9 + "/module1:grad"), [(c1, "conv", slice(None))], extra_fn=g_join, n_terms=2)
    fs.drop(c2 + ":z", c2 + ":dz", c1 + ":z", c1 + ":dz", c1 + ":grad", c1, m9 + "/module1:grad")

    # B: conv1 (8 -> 8, ReLU), x = the last unit's output; its gradient comes from conv2 alone
    x_b = lambda n, lo, hi: out[n, lo:hi].astype(np.float64)
    a1 = stored_or_bn(U + "conv1", True)
    gc1 = fs.t(U + "conv1:grad")
    fs.check_stats(U + "conv1")
    fs.check_forward(U + "conv1", "conv", x_b)
    fs.check_data_gradient(U + "conv1:grad", gc1, [(U + "conv2", "conv", slice(None))])
    fs.check_bn_backward(U + "conv1", lambda n, a, b: gc1[n, a:b].astype(np.float64) * (a1(n, a, b) > 0))
    # (its weight gradient runs the same kernel instantiation as module2/resnet_conv1's above: not repeated at full size)
    fs.check_data_gradient(m9 + "/module2:grad", gout, [(U + "conv1", "conv", slice(None))])
    fs.drop(U + "conv1:z", U + "conv1:dz", U + "conv1:grad", U + "conv1", U + "conv2:dz", m9 + "/module2", m9 + "/module2:grad")

    # C: the first decoder unit of level 0: input = tf.concat([deconv, conv0]) (16 channels), 3x3 16 -> 8 + 1x1 shortcut
    c1, sc = m9 + "/module1/resnet_conv1", m9 + "/module1/shortcut"
    xd, x0 = fs.t(dec), fs.t(U + "conv0")
    x_c = lambda n, lo, hi: np.concatenate([xd[n, lo:hi], x0[n, lo:hi]], axis=-1).astype(np.float64)
    fs.check_stats(c1)
    fs.check_forward(c1, "conv", x_c)
    fs.check_weight_gradient(c1, "conv", x_c)
    F = xd.shape[-1]
    terms = lambda s: [(c1, "conv", s), (sc, "conv", s)]
    fs.check_data_gradient(dec + ":grad", fs.t(dec + ":grad"), terms(slice(0, F)), n_terms=2)
    if bf16 and F == 8:   # the skip's share of the concat gradient is its own tensor ...
        try:
            g2 = net.debug_tensor(U + "conv0:grad2")
        except Exception as e:   # ... until the encoder's data gradients accumulate into it (default): by the end of the backward
            # pass it holds the whole d(conv0 activation); the sum over ALL consumers is checked on the small nets
            # (_insitu.InSitu.total_grad) and the switch case URSN_BF16_SKIP0_MERGE=0 keeps the separate tensor
            assert "no second gradient tensor" in str(e), e
            g2 = None
        if g2 is not None:
            fs.check_data_gradient(U + "conv0:grad2", g2, terms(slice(F, 2 * F)), n_terms=2)
    gd = fs.t(dec + ":grad")
    fs.check_bn_backward(dec, lambda n, a, b: gd[n, a:b].astype(np.float64) * (xd[n, a:b] > 0))
    fs.drop(c1 + ":z", c1 + ":dz", sc + ":dz", U + "conv0", dec, dec + ":grad")

    # D + E: level 1 -- 16 -> 16 behind an identity shortcut, and the transposed conv that leaves the level
    c1, c2 = m8 + "/module2/resnet_conv1", m8 + "/module2/resnet_conv2"
    x_d = stored_or_bn(m8 + "/module1", False)
    out8, gout8 = fs.t(m8 + "/module2"), fs.t(m8 + "/module2:grad")
    g_join8 = lambda n, a, b: gout8[n, a:b].astype(np.float64) * (out8[n, a:b] > 0)
    fs.check_stats(c1)
    fs.check_forward(c1, "conv", x_d)
    fs.check_weight_gradient(c1, "conv", x_d)
    fs.check_data_gradient(m8 + "/module1:grad", fs.t(m8 + "/module1:grad"), [(c1, "conv", slice(None))], extra_fn=g_join8, n_terms=2)
    fs.check_bn_backward(c2, g_join8)
    x_e = lambda n, lo, hi: out8[n, lo:hi].astype(np.float64)
    fs.check_stats(dec)
    fs.check_forward(dec, "deconv", x_e)
    fs.check_weight_gradient(dec, "deconv", x_e)
    fs.check_data_gradient(m8 + "/module2:grad", gout8, [(dec, "deconv", slice(None))])
    fs.tol.report("full size")
    print("full-size check seconds:", {k: round(v, 1) for k, v in sorted(fs.times.items(), key=lambda kv: -kv[1])})
    assert set(fs.tol.worst) >= {"z", "rstd", "dz", "dw", "dx"}


def _full_size_step(dims, base, ncls, N, prec):
    from uresnet_amd import synthetic_io as sio
    P = as_f32_exact(oracle_params(dims, base, ncls))
    b = [sio.lartpc_sparse(dims, ncls, i) for i in range(N)]
    data, label, weight = (np.stack([x[j] for x in b]) for j in range(3))
    weight /= weight.sum(axis=1, keepdims=True)
    net = uresnet(dims=list(dims), num_class=ncls, base_num_outputs=base)
    net.construct(trainable=True, use_weight=True, learning_rate=1e-3, precision=prec)
    net.set_variables(P)
    net.zero_gradients(None)
    res, _ = net.accum_gradients(None, data, label, weight)
    assert np.isfinite(res[1])
    return net, P


def test_cfg3_full_size_level0_and_level1_layers_fp32():
    """BASELINE configs[2] as itself: 3-D 192^3 x 1, F = 8, 3 classes, batch 4, fp32 (the bench workload)."""
    net, P = _full_size_step((192, 192, 192, 1), 8, 3, 4, "fp32")
    _full_size_checks(net, P, False)


def test_cfg5_full_size_level0_and_level1_layers_bf16():
    """BASELINE configs[4]: 3-D 256^3 bf16 (batch 2 of the bench's 4 to bound the oracle's time): tensors above 2^31 bytes,
    XCD remap at 262,144 workgroups, z-segment splitting, one-workgroup-per-CU LDS budgets."""
    net, P = _full_size_step((256, 256, 256, 1), 8, 3, 2, "bf16")
    _full_size_checks(net, P, True)
