"""Run-to-run bitwise reproducibility of one training step (fixed-order reductions, two streams): two nets, same seed
and batch, gradients and post-Adam weights must be identical bit for bit.  python tools/determinism_check.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import uresnet_amd  # noqa: F401,E402
from uresnet_amd import uresnet  # noqa: E402
from uresnet_amd import synthetic_io as sio  # noqa: E402

dims = (96, 96, 96, 1)
vol = [sio.lartpc_sparse(dims, 3, i) for i in range(2)]
data = np.stack([v[0] for v in vol]); label = np.stack([v[1] for v in vol]); weight = np.stack([v[2] for v in vol])
weight /= weight.sum(axis=1, keepdims=True)
out = []
for rep in range(2):
    net = uresnet(dims=list(dims), num_class=3, base_num_outputs=8)
    net.construct(trainable=True, use_weight=True, learning_rate=1e-3, seed=7)
    for _ in range(2):
        net.zero_gradients(None)
        res, _ = net.accum_gradients(None, data, label, weight)
        net.apply_gradients(None)
    out.append((res[1], net.get_gradients(), net.get_variables()))
assert out[0][0] == out[1][0], (out[0][0], out[1][0])
for k in out[0][1]:
    assert np.array_equal(out[0][1][k], out[1][1][k]), "gradient differs: " + k
for k in out[0][2]:
    assert np.array_equal(out[0][2][k], out[1][2][k]), "weight differs: " + k
print("bitwise identical: loss", out[0][0], "over", len(out[0][1]), "gradient tensors")
