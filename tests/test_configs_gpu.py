"""Every BASELINE.json config exercised AS ITSELF: the config's exact model tuple
(ndim, F, classes, depth 5, USE_WEIGHTS) against the fp64 numpy oracle at a spatial size the oracle
affords, and at the config's FULL size through size-independent properties (finite loss, softmax
rows sum to 1, accuracies in [0,1], bitwise run-to-run determinism, specialised kernels == the
generic gather kernels).  PARITY UNPINNED (oracle/__init__.py): the oracle is a restatement.

    cfg1  config/train2d.cfg: 2-D 256x256x1 -> 3 classes, F=16, batch 4, USE_WEIGHTS False  (exact size)
    cfg2  2-D 512x512x1 -> 5 classes, F=16, batch 16           (oracle leg at 128x128 batch 4)
    cfg3  3-D 192^3x1 -> 3 classes, F=8, batch 4               (oracle legs at 64^3 batch 2 and 128^3 batch 1)
    lib/uresnet.py:128-130 smoke shapes 512^2 and 128^3 (defaults F=16) ride in the same tests.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import uresnet_np as O
from _net import as_f32_exact, fp32_noise_floor, l2_rel, make_inputs, max_rel, oracle_params, parallel_oracle
from uresnet_amd import uresnet

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _forward_checks(net, data, m, res, tag):
    """north_star tolerances: logits/softmax within 1e-3 relative, labels bit-exact away from near-ties.
    Asserts what the path achieves (an order of magnitude inside the 1e-3 ceiling) and prints it."""
    zl = net.debug_tensor("UResNet/conv2:z")
    e_z = max_rel(zl, m["acts"]["UResNet/conv2:z"])
    sm = net.inference(None, data)[0]
    ref = m["softmax"]
    e_sm_abs = float(np.abs(sm - ref).max())
    big = ref > 1e-3
    e_sm_rel = float((np.abs(sm - ref)[big] / ref[big]).max())     # element-wise relative, where p > 1e-3
    srt = np.sort(m["logits"], axis=-1)
    safe = (srt[..., -1] - srt[..., -2]) > 1e-3
    print("%s: conv2:z max-rel %.2e, softmax max-abs %.2e, softmax element-wise rel (p>1e-3) %.2e, safe pixels %.4f"
          % (tag, e_z, e_sm_abs, e_sm_rel, safe.mean()))
    assert e_z < 2e-4, e_z
    assert e_sm_abs < 1e-4, e_sm_abs
    assert e_sm_rel < 1e-3, e_sm_rel
    assert np.allclose(sm.sum(-1), 1.0, atol=1e-5)
    assert safe.mean() > 0.95
    assert np.array_equal(sm.argmax(-1)[safe], m["pred"][safe])
    n_unsafe = int((~safe).sum())
    assert abs(res[2] - m["acc_all"]) <= (n_unsafe + 0.5) / safe.size
    assert abs(res[1] - m["loss"]) <= 1e-4 * abs(m["loss"])


def _grad_errors(g, g_ref):
    return {k: l2_rel(g[k], g_ref[k]) for k in g_ref if np.abs(g_ref[k]).max() > 1e-12}


CONFIG_CASES = [
    # tag, dims, F, classes, batch, use_weight, tight gradient bound (None: relative to the fp32 noise floor)
    ("cfg1_exact_2d256_f16_c3_b4", (256, 256, 1), 16, 3, 4, False, None),
    ("cfg2_model_2d128_f16_c5_b4", (128, 128, 1), 16, 5, 4, True, None),
    ("cfg3_model_3d64_f8_c3_b2", (64, 64, 64, 1), 8, 3, 2, True, None),
    # well conditioned at full depth: the bottleneck BatchNorm sees 4^3 = 64 samples per channel
    ("cfg3_model_3d128_f8_c3_b1_wellcond", (128, 128, 128, 1), 8, 3, 1, True, 2e-3),
]


@pytest.mark.parametrize("case", CONFIG_CASES, ids=[c[0] for c in CONFIG_CASES])
def test_baseline_config_model_against_oracle(case):
    tag, dims, base, ncls, N, use_w, tight = case
    P = as_f32_exact(oracle_params(dims, base, ncls))
    data, label, weight = make_inputs(dims, ncls, N, seed=17)
    w = weight if use_w else None
    with parallel_oracle():   # the oracle's own conv functions, slab-parallel (tests/_net.py)
        g_ref, m = O.step_gradients(P, dims, base, data, label, w, keep_acts=True)
    net = uresnet(dims=list(dims), num_class=ncls, base_num_outputs=base)   # num_strides = 5 as in the reference
    net.construct(trainable=True, use_weight=use_w, learning_rate=1e-3)
    assert net._n_params == {(2, 16, 3): 16858979, (2, 16, 5): 16859269, (3, 8, 3): 12468083}[(len(dims) - 1, base, ncls)]
    net.set_variables(P)
    net.zero_gradients(None)
    res, _ = net.accum_gradients(None, data, label, w)
    _forward_checks(net, data, m, res, tag)
    for name in ["UResNet/conv0", "UResNet/resnet_module2/module1", "UResNet/resnet_module4/module2", "UResNet/deconv0",
                 "UResNet/resnet_module7/module1", "UResNet/resnet_module9/module2", "UResNet/conv1"]:
        assert max_rel(net.debug_tensor(name), m["acts"][name]) < 1e-3, name
    errs = _grad_errors(net.get_gradients(), g_ref)
    worst_w = sorted(((k, e) for k, e in errs.items() if k.endswith("/weights")), key=lambda kv: -kv[1])[:3]
    worst_b = sorted(((k, e) for k, e in errs.items() if k.endswith("/beta")), key=lambda kv: -kv[1])[:3]
    print("%s: worst gradient relative L2: weights %s, beta %s" % (tag, [(k, "%.2e" % e) for k, e in worst_w],
                                                                     [(k, "%.2e" % e) for k, e in worst_b]))
    g32, _ = fp32_noise_floor(P, dims, base, data, label, w)
    floor = {k: l2_rel(g32[k], g_ref[k]) for k in errs}
    if tight is not None:
        # Full depth, 64 samples per channel at the bottleneck BatchNorm.  Measured (profiles/r02_gputest.log): filter
        # gradients sit at a median 3.8e-3 relative L2 from the fp64 oracle (10 of 58 tensors within 2e-3) -- an INDEPENDENT
        # fp32 evaluation of the oracle (numpy float32, plain summation) sits at 4.7e-2, twelve times farther: the 2e-3 that
        # kernel-level parity reaches (tests/test_ops_gpu.py: 2e-5) is not available to ANY fp32 evaluation of this 58-layer
        # batch-statistics network at 128^3 x 1; fp64 BatchNorm sums and fixed-order reductions buy the factor twelve.
        # Asserted: median <= 5e-3, every filter gradient <= 1e-2 and <= half the fp32 floor of that tensor.
        # d(beta) = sum(g) is held to 1e-2: a constant added to a BatchNorm output is removed again by the next BatchNorm
        # except through zero-padded borders and ReLU kinks, so these sums cancel to ~1e-3 of sum|g| (fp64 oracle:
        # |sum g| / sum|g| = 1e-3..4e-3 for resnet_conv1's beta) and carry the rounding of the terms.
        wk = [k for k in errs if k.endswith("/weights")]
        print("%s: filter gradients: median error %.2e (fp32 floor %.2e), within 2e-3: %d of %d; worst error %.2e, worst error / floor %.2f"
              % (tag, np.median([errs[k] for k in wk]), np.median([floor[k] for k in wk]),
                 sum(errs[k] <= tight for k in wk), len(wk), max(errs[k] for k in wk), max(errs[k] / max(floor[k], 1e-9) for k in wk)))
        assert np.median([errs[k] for k in wk]) <= 5e-3
        bad = [(k, errs[k], floor[k]) for k in wk if errs[k] > 1e-2 or errs[k] > 0.5 * floor[k]]
        assert not bad, bad
        assert worst_b[0][1] <= 1e-2, worst_b
    else:
        bad = [(k, e, floor[k]) for k, e in errs.items() if e > min(max(2e-3, 4 * floor[k]), 5e-2)]
        assert not bad, bad


# ---- full-size property checks -----------------------------------------------------------------------------------
FULL = {
    "cfg2_2d512_f16_c5_b16": ((512, 512, 1), 16, 5, 16),
    "cfg3_3d192_f8_c3_b4": ((192, 192, 192, 1), 8, 3, 4),
    "ref_smoke_2d512_f16_c3_b1": ((512, 512, 1), 16, 3, 1),      # lib/uresnet.py:128,132-137
    "ref_smoke_3d128_f16_c3_b1": ((128, 128, 128, 1), 16, 3, 1),  # lib/uresnet.py:129-130
}

_CHILD = r"""
import sys, json, hashlib
sys.path.insert(0, %(root)r)
import numpy as np
import uresnet_amd
from uresnet_amd import uresnet
from uresnet_amd import synthetic_io as sio
dims, base, ncls, N = %(case)r
net = uresnet(dims=list(dims), num_class=ncls, base_num_outputs=base)
net.construct(trainable=True, use_weight=True, learning_rate=1e-3, seed=99)
b = [sio.lartpc_sparse(dims, ncls, i) for i in range(N)]
data, label, weight = (np.stack([x[j] for x in b]) for j in range(3))
weight /= weight.sum(axis=1, keepdims=True)
out = {}
for rep in range(2):
    net.zero_gradients(None)
    res, _ = net.accum_gradients(None, data, label, weight)
    g = net.get_gradients()
    h = hashlib.sha256()
    for k in sorted(g):
        h.update(np.ascontiguousarray(g[k]).tobytes())
    out.setdefault("loss", []).append(res[1]); out.setdefault("acc", []).append(res[2:]); out.setdefault("hash", []).append(h.hexdigest())
sm = net.inference(None, data[:1])[0]
out["softmax_rowsum_err"] = float(np.abs(sm.sum(-1) - 1.0).max())
out["softmax_min"], out["softmax_max"] = float(sm.min()), float(sm.max())
np.savez(%(npz)r, **{k.replace("/", "|"): v for k, v in g.items()})
print("RESULT " + json.dumps(out))
"""


def _run_child(case, npz, env):
    code = _CHILD % {"root": ROOT, "case": case, "npz": npz}
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=1100)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [x for x in p.stdout.split("\n") if x.startswith("RESULT ")][-1]
    return json.loads(line[7:])


@pytest.mark.parametrize("tag", sorted(FULL))
def test_full_size_properties_and_dispatch_consistency(tag, tmp_path):
    """The auto-dispatched plan at the config's real size (z-segment planner, XCD tile order, multi-round implicit-GEMM
    grids, two-stage slab reductions, second stream) against the generic gather kernels on the same weights and batch."""
    case = FULL[tag]
    fast = _run_child(case, str(tmp_path / "fast.npz"), {})
    assert all(np.isfinite(fast["loss"])) and fast["loss"][0] > 0
    assert fast["loss"][0] == fast["loss"][1] and fast["hash"][0] == fast["hash"][1], "step is not bitwise reproducible"
    for acc in fast["acc"]:
        assert 0.0 <= acc[0] <= 1.0 and 0.0 <= acc[1] <= 1.0
    assert fast["softmax_rowsum_err"] < 1e-5 and fast["softmax_min"] >= 0.0 and fast["softmax_max"] <= 1.0
    gen = _run_child(case, str(tmp_path / "generic.npz"), {"URSN_DISABLE_TILED": "1", "URSN_WGRAD_STREAM": "0"})
    assert abs(gen["loss"][0] - fast["loss"][0]) < 1e-5 * abs(gen["loss"][0])
    a, b = np.load(str(tmp_path / "generic.npz")), np.load(str(tmp_path / "fast.npz"))
    errs = sorted(((l2_rel(b[k], a[k]), k) for k in a.files if np.linalg.norm(a[k]) > 0), reverse=True)
    print("%s: loss %.6f, worst specialised-vs-generic gradient relative L2 %s" % (tag, fast["loss"][0], errs[:3]))
    # Two fp32 evaluations with different summation orders.  Which side of a ReLU / argmax boundary a handful of values
    # land on moves every upstream gradient a little: switching ONE kernel family (tools/ab_env.py: URSN_STRIDE2=0 or
    # URSN_SCATTER_LDS=0) moves the median filter gradient of cfg2 by 6e-3 against the generic path while the loss agrees to
    # all printed digits and the fused statistics are exact to 4e-8 (tools/stats_check.py); d(beta) of resnet_conv1 (a
    # cancelling sum) moves by up to 1.7e-2.  Bounds: worst 3e-2, median 1e-2.
    assert errs[0][0] < 3e-2, errs[:3]
    assert np.median([e for e, _ in errs]) < 1e-2
