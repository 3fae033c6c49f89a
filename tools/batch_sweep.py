"""Plans the cfg3 net for batch 4, then feeds batches 4, 1, 3, 2 (scratch sized at plan time must cover the per-call
launch geometry of every smaller batch).  Run on a GPU box: python tools/batch_sweep.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import uresnet_amd  # noqa: F401,E402
from uresnet_amd import uresnet  # noqa: E402
from uresnet_amd import synthetic_io as sio  # noqa: E402

dims = (192, 192, 192, 1)
net = uresnet(dims=list(dims), num_class=3, base_num_outputs=8)
net.construct(trainable=True, use_weight=True, learning_rate=1e-3, seed=5)
vol = [sio.lartpc_sparse(dims, 3, i) for i in range(4)]
data = np.stack([v[0] for v in vol]); label = np.stack([v[1] for v in vol]); weight = np.stack([v[2] for v in vol])
weight /= weight.sum(axis=1, keepdims=True)
for nb in (4, 1, 3, 2):
    net.zero_gradients(None)
    res, _ = net.accum_gradients(None, data[:nb], label[:nb], weight[:nb])
    net.apply_gradients(None)
    torch.cuda.synchronize()
    print("batch", nb, "loss", res[1], "acc", res[2], res[3], flush=True)
    assert np.isfinite(res[1])
print("ok")
