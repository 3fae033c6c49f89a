"""Inference throughput of the ana path (config/ana3d.cfg shape): softmax volume and on-device label rule, device-resident
input.  Run on a GPU box: python tools/infer_bench.py [batch]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import uresnet_amd  # noqa: F401,E402
from uresnet_amd import uresnet  # noqa: E402
from uresnet_amd import synthetic_io as sio  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dims = (192, 192, 192, 1)
net = uresnet(dims=list(dims), num_class=3, base_num_outputs=8)
net.construct(trainable=False, use_weight=False, seed=5)
data = torch.from_numpy(np.stack([sio.lartpc_sparse(dims, 3, i)[0] for i in range(batch)])).cuda()
for name, fn in (("softmax", lambda: net.inference(None, data, as_numpy=False)),
                 ("labels", lambda: net.inference_labels(None, data, as_numpy=False))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print("%-8s batch %d: %.2f ms -> %.1f images/s" % (name, batch, ms, batch / ms * 1e3))
