"""CPU tests of the oracle itself: golden fixtures, the two independent formulations against each
other, and the pinned structural facts (shapes, parameter counts, TF-SAME alignment, TF-form Adam).
PARITY UNPINNED: see oracle/__init__.py -- these pin the restatement, not TensorFlow."""
import os

import numpy as np
import pytest
import torch

from oracle import uresnet_np as O
from oracle import uresnet_torch as T

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False) as f:
        return {k: f[k] for k in f.files}


@pytest.mark.parametrize("name", ["net2d_32x32_f4_ns3", "net3d_16x16x16_f4_ns2"])
def test_numpy_oracle_reproduces_golden(name):
    G = load(name)
    dims, base, ns = tuple(int(d) for d in G["dims"]), int(G["base"]), int(G["num_strides"])
    P = O.OrderedDict((k[6:], G[k].astype(np.float64)) for k in G if k.startswith("param:"))
    w = G["weight"] if int(G["use_weight"]) else None
    g, m = O.step_gradients(P, dims, base, G["data"], G["label"], w, num_strides=ns)
    assert abs(m["loss"] - float(G["loss"])) < 1e-9 * abs(float(G["loss"]))
    assert m["acc_all"] == pytest.approx(float(G["acc_all"]), abs=1e-12)
    assert m["acc_nonzero"] == pytest.approx(float(G["acc_nonzero"]), abs=1e-12)
    assert np.abs(m["logits"] - G["logits"]).max() < 1e-5
    for k in P:
        assert np.abs(g[k] - G["grad:" + k]).max() <= 2e-6 * (np.abs(G["grad:" + k]).max() + 1e-30), k


@pytest.mark.parametrize("name", ["net2d_32x32_f4_ns3", "net3d_16x16x16_f4_ns2"])
def test_torch_oracle_reproduces_golden_and_adam(name):
    G = load(name)
    dims, base, ns = tuple(int(d) for d in G["dims"]), int(G["base"]), int(G["num_strides"])
    P = {k[6:]: torch.tensor(G[k].astype(np.float64), requires_grad=True) for k in G if k.startswith("param:")}
    use_w = bool(int(G["use_weight"]))
    opt = T.Adam(P, lr=1e-3)
    losses = []
    for it in range(2):
        g, m = T.step_gradients(P, dims, base, G["data"], G["label"], G["weight"] if use_w else None, num_strides=ns)
        if it == 0:
            for k in P:
                assert np.abs(g[k].numpy() - G["grad:" + k]).max() <= 2e-6 * (np.abs(G["grad:" + k]).max() + 1e-30), k
        losses.append(m["loss"])
        opt.apply(P, g)
    assert np.allclose(losses, G["adam_losses"], rtol=1e-9)
    for k in P:
        assert np.abs(P[k].detach().numpy() - G["adam2:" + k]).max() < 2e-7, k


def test_ops_golden():
    G = load("ops")
    for tag in ["c3s1", "c3s2", "c1s2", "c2s2odd"]:
        s = int(G[tag + ":stride"])
        y = O.conv_fwd(G[tag + ":x"], G[tag + ":w"], s)
        dx, dw = O.conv_bwd(G[tag + ":x"], G[tag + ":w"], s, G[tag + ":dy"])
        assert np.allclose(y, G[tag + ":y"]) and np.allclose(dx, G[tag + ":dx"]) and np.allclose(dw, G[tag + ":dw"])
        xt = torch.tensor(G[tag + ":x"], requires_grad=True)
        wt = torch.tensor(G[tag + ":w"], requires_grad=True)
        yt = T._to_nxc(T.conv_same(T._to_ncx(xt), wt, s))
        assert np.allclose(yt.detach().numpy(), y)
        gx, gw = torch.autograd.grad(yt, [xt, wt], torch.tensor(G[tag + ":dy"]))
        assert np.allclose(gx.numpy(), dx) and np.allclose(gw.numpy(), dw)
    for tag in ["d3", "d2"]:
        y = O.deconv_fwd(G[tag + ":x"], G[tag + ":w"])
        dx, dw = O.deconv_bwd(G[tag + ":x"], G[tag + ":w"], G[tag + ":dy"])
        assert np.allclose(y, G[tag + ":y"]) and np.allclose(dx, G[tag + ":dx"]) and np.allclose(dw, G[tag + ":dw"])
        xt = torch.tensor(G[tag + ":x"], requires_grad=True)
        wt = torch.tensor(G[tag + ":w"], requires_grad=True)
        yt = T._to_nxc(T.deconv_same(T._to_ncx(xt), wt))
        assert np.allclose(yt.detach().numpy(), y)
        gx, gw = torch.autograd.grad(yt, [xt, wt], torch.tensor(G[tag + ":dy"]))
        assert np.allclose(gx.numpy(), dx) and np.allclose(gw.numpy(), dw)


def test_tf_same_alignment():
    """k3 s2 on even size pads 0 before / 1 after; the transposed conv is the exact
    adjoint of that stride-2 conv and keeps outputs [0, 2*in)."""
    x = np.arange(8, dtype=np.float64).reshape(1, 8, 1)  # 1-D would need ndim 1; use 2-D with unit axis
    x2 = x.reshape(1, 1, 8, 1)
    w = np.zeros((3, 3, 1, 1)); w[1, 0, 0, 0] = 1.0   # picks x[2o + 0] along the last axis, centre row
    y = O.conv_fwd(np.repeat(x2, 2, axis=1), w, 2)
    assert y.shape == (1, 1, 4, 1)
    rng = np.random.default_rng(0)
    xx = rng.standard_normal((2, 6, 8, 3)); ww = rng.standard_normal((3, 3, 3, 5)); g = rng.standard_normal((2, 3, 4, 5))
    lhs = (O.conv_fwd(xx, ww, 2) * g).sum()
    wd = np.transpose(ww, (0, 1, 3, 2))      # deconv filter layout [k,k,Cout,Cin] with Cout<->Cin swapped roles
    rhs = (xx * O.deconv_fwd(g, np.transpose(wd, (0, 1, 3, 2)))).sum()
    assert abs(lhs - rhs) < 1e-9 * abs(lhs)
    dx, _ = O.conv_bwd(xx, ww, 2, g)
    assert np.allclose(dx, O.deconv_fwd(g, ww))


def test_parameter_counts_match_survey_appendix_a():
    count = lambda nd, F, c: sum(int(np.prod(s)) for _, s in O.param_specs(nd, 1, F, c))
    assert count(3, 8, 3) == 12468083
    assert count(2, 16, 3) == 16858979
    assert count(2, 16, 5) == 16859269
    assert len(O.layer_table(3, 1, 8, 3)) == 58
    names = [n for n, _ in O.param_specs(3, 1, 8, 3)]
    assert names[0] == "UResNet/conv0/weights" and names[1] == "UResNet/conv0/BatchNorm/beta"
    assert "UResNet/resnet_module5/module1/shortcut/weights" in names
    assert "UResNet/resnet_module0/module2/shortcut/weights" not in names


def test_batchnorm_semantics():
    rng = np.random.default_rng(1)
    z = rng.standard_normal((3, 5, 7, 4)) * 3 + 2
    beta = rng.standard_normal(4)
    y, (xhat, r) = O.bn_fwd(z, beta)
    assert np.allclose((y - beta).mean(axis=(0, 1, 2)), 0, atol=1e-12)
    assert np.allclose(r, 1 / np.sqrt(z.var(axis=(0, 1, 2)) + 1e-3))
    dy = rng.standard_normal(z.shape)
    dz, dbeta = O.bn_bwd((xhat, r), dy)
    eps = 1e-6
    i = (1, 2, 3, 1)
    zp = z.copy(); zp[i] += eps
    num = ((O.bn_fwd(zp, beta)[0] - y) * dy).sum() / eps
    assert abs(num - dz[i]) < 1e-5
    assert np.allclose(dbeta, dy.sum(axis=(0, 1, 2)))


def test_adam_is_tf_form():
    P = {"a": np.array([1.0, -2.0])}
    opt = O.Adam(P, lr=0.1)
    opt.apply(P, {"a": np.array([0.5, -4.0])})
    lr_t = 0.1 * np.sqrt(1 - 0.999) / (1 - 0.9)
    m, v = 0.1 * np.array([0.5, -4.0]), 0.001 * np.array([0.25, 16.0])
    assert np.allclose(P["a"], np.array([1.0, -2.0]) - lr_t * m / (np.sqrt(v) + 1e-8))


def test_loss_metrics_and_ana_rule():
    logits = np.array([[[2.0, 2.0, 1.0], [0.0, 1.0, 3.0]]])       # tie -> lowest index
    data = np.array([[0.0, 2.0]]); label = np.array([[0.0, 1.0]]); weight = np.array([[0.25, 0.75]])
    m = O.loss_and_metrics(logits, data, label, weight)
    assert m["pred"].tolist() == [[0, 2]] and m["acc_all"] == 0.5 and m["acc_nonzero"] == 0.0
    ce = -np.log(O.softmax(logits))[0, [0, 1], [0, 1]]
    assert m["loss"] == pytest.approx((ce * weight[0]).sum())
    assert np.isnan(O.loss_and_metrics(logits, np.zeros((1, 2)), label)["acc_nonzero"])
    sm = np.array([[0.2, 0.5, 0.3], [0.2, 0.3, 0.5], [0.2, 0.4, 0.4]])
    assert O.ana_label_rule(sm, np.array([5.0, 5.0, 0.5])).tolist() == [1.0, 2.0, 0.0]
    assert O.ana_label_rule(sm, np.array([5.0, 5.0, 5.0])).tolist() == [1.0, 2.0, 2.0]


def test_gradient_accumulation_is_a_sum_not_a_mean():
    dims, base, ncls, ns = (16, 16, 1), 4, 3, 2
    P = O.init_params(2, 1, base, ncls, seed=3, num_strides=ns)
    rng = np.random.default_rng(5)
    mbs = []
    for _ in range(2):
        d = rng.random((2, 256)); l = rng.integers(0, 3, (2, 256)).astype(float); w = np.full((2, 256), 1 / 256.)
        mbs.append((d, l, w))
    g0, _ = O.step_gradients(P, dims, base, *mbs[0], num_strides=ns)
    g1, _ = O.step_gradients(P, dims, base, *mbs[1], num_strides=ns)
    P2 = type(P)((k, v.copy()) for k, v in P.items())
    _, acc = O.train_step(P2, O.Adam(P2), dims, base, mbs, num_strides=ns)
    for k in P:
        assert np.allclose(acc[k], g0[k] + g1[k])


# ---- the one reference-held artefact: the saved GraphDef (tests/golden/ref_graph.json, made by make_ref_graph.py) ----
def _ref_graph():
    import json
    with open(os.path.join(GOLD, "ref_graph.json")) as f:
        return json.load(f)


def test_oracle_topology_against_reference_graph():
    """What the reference's saved graph (an older revision: 2-D 512x512, F=16, 3 classes) pins for
    the current code: 53 convs + 5 transposed convs in this order with these strides / SAME / NHWC, the filter shapes
    (k and channels; NOT conv0/conv1's 7x7 kernel), deconv filters stored [k,k,Cout,Cin], tf.concat([deconv_i, skip])
    with skip = conv0 / resnet_module{0..3}/module2, Xavier-uniform bounds, BatchNorm = beta + moving averages (no
    gamma), two-pass moments over [N,H,W].  Numerical parity stays UNPINNED: a GraphDef holds no activations."""
    G = _ref_graph()
    h = G["op_histogram"]
    assert (h["Conv2D"], h["Conv2DBackpropInput"], h["Conv2DBackpropFilter"]) == (58, 58, 58)
    assert G["graph_producer_version"] == 24 and G["num_nodes"] == 10303
    L = O.layer_table(2, 1, 16, 3)
    assert len(L) == len(G["forward_convs"]) == 58
    old_kernel7 = {"UResNet/conv0", "UResNet/conv1"}          # the saved revision used 7x7 there (lib/uresnet.py:39,106 now 3)
    for l, c in zip(L, G["forward_convs"]):
        assert l["name"] == c["scope"]
        assert (l["kind"] == "deconv") == (c["op"] == "Conv2DBackpropInput"), l["name"]
        assert c["padding"] == "SAME" and c["data_format"] == "NHWC"
        assert c["strides"] == [1, l["stride"], l["stride"], 1], l["name"]
        if l["name"] in old_kernel7:
            assert c["filter_shape"][:2] == [7, 7] and c["filter_shape"][2:] == l["wshape"][2:]
        else:
            assert c["filter_shape"] == l["wshape"], l["name"]    # deconv: [k,k,Cout,Cin]
    d0 = [c for c in G["forward_convs"] if c["scope"] == "UResNet/deconv0"][0]
    assert d0["filter_shape"] == [3, 3, 256, 512]
    # the input gradient of a transposed conv is a plain stride-2 Conv2D with the same filter (Appendix B-2)
    assert G["deconv0_input_gradient_ops"] == ["Conv2D", "Conv2DBackpropFilter"]
    want_skip = ["UResNet/resnet_module3/module2", "UResNet/resnet_module2/module2", "UResNet/resnet_module1/module2",
                 "UResNet/resnet_module0/module2", "UResNet/conv0"]
    assert [c["inputs"] for c in G["concats"]] == [["UResNet/deconv%d" % i, want_skip[i]] for i in range(5)]
    # Xavier uniform: limit = sqrt(6 / (fan_in + fan_out)), fan = k^d * C  (Appendix B-6)
    for scope, x in G["xavier_limits"].items():
        sh = x["shape"]
        fan = sh[0] * sh[1]
        assert abs(x["limit"] - np.sqrt(6.0 / (fan * (sh[2] + sh[3])))) < 1e-8, scope
    P = O.init_params(2, 1, 16, 3, seed=7)
    for l in L:
        if l["name"] in old_kernel7:
            continue
        w, lim = P[l["name"] + "/weights"], G["xavier_limits"][l["name"]]["limit"]
        assert np.abs(w).max() <= lim * (1 + 1e-7), l["name"]
        if w.size >= 4096:
            assert np.abs(w).max() > 0.99 * lim, l["name"]
        assert np.all(P[l["name"] + "/BatchNorm/beta"] == 0)
    assert G["batchnorm_variable_sets"] == [["beta", "moving_mean", "moving_variance"]]      # no gamma (B-3a)
    assert G["moments"] == {"has_squared_difference": True, "has_stop_gradient_on_mean": True, "mean_reduction_axes": [0, 1, 2]}
    assert G["moving_average_decay_constants"] == [0.001]                                    # 1 - 0.999 (B-3)
    assert [p["shape"] for p in G["placeholders"]] == [[-1, 262144]] * 3                      # flat [N, 512*512] feeds


def test_fast_stride1_conv_helpers_are_the_oracle_primitives():
    """tests/_insitu.py evaluates the oracle's stride-1 3^d convolution and its two gradients on flattened padded tensors (the
    full-size in-situ checks spend their time there); the three helpers must BE oracle.conv_fwd / conv_bwd: same terms,
    fp64 rounding only (2-D and 3-D, ragged sizes, one-voxel axes)."""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _insitu import fast_conv_dw, fast_conv_dx, fast_conv_fwd
    rng = np.random.default_rng(5)
    for S, ci, co in (((5, 6, 7), 3, 4), ((6, 9), 5, 2), ((1, 4, 3), 2, 2), ((8, 8, 8), 8, 8)):
        x = rng.standard_normal((2,) + S + (ci,))
        w = rng.standard_normal((3,) * len(S) + (ci, co))
        dy = rng.standard_normal((2,) + S + (co,))
        dx, dw = O.conv_bwd(x, w, 1, dy)
        y = O.conv_fwd(x, w, 1)
        assert np.abs(fast_conv_fwd(x, w) - y).max() <= 1e-12 * np.abs(y).max()
        assert np.abs(fast_conv_dx(dy, w) - dx).max() <= 1e-12 * np.abs(dx).max()
        assert np.abs(fast_conv_dw(x, dy) - dw).max() <= 1e-12 * np.abs(dw).max()


def test_slab_parallel_oracle_is_the_oracle():
    """tests/_net.py::parallel_oracle runs oracle.conv_fwd / conv_bwd per slab of the first spatial axis in threads (the 128^3
    oracle steps of the GPU suite): same values as the plain functions, 3-D and 2-D, slab counts that do and do not divide."""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _net import parallel_oracle
    rng = np.random.default_rng(6)
    for S, ci, co in (((21, 40, 40), 4, 3), ((64, 96), 8, 5)):
        x = rng.standard_normal((2,) + S + (ci,))
        w = rng.standard_normal((3,) * len(S) + (ci, co))
        dy = rng.standard_normal((2,) + S + (co,))
        y0 = O.conv_fwd(x, w, 1)
        dx0, dw0 = O.conv_bwd(x, w, 1, dy)
        for rows in (None, 5, 16):
            with parallel_oracle(workers=4, min_rows=8, rows=rows):
                y1 = O.conv_fwd(x, w, 1)
                dx1, dw1 = O.conv_bwd(x, w, 1, dy)
                assert O.conv_fwd(x[:, :4], w, 1).shape[1] == 4          # small tensors take the plain path
            assert np.array_equal(y1, y0) and np.array_equal(dx1, dx0)
            assert np.abs(dw1 - dw0).max() <= 1e-12 * np.abs(dw0).max()
    assert O.conv_fwd.__module__ == "oracle.uresnet_np"                  # restored
