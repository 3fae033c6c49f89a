"""Is a workload bound by the host's launch rate?  Enqueues K steps without synchronising and compares the host time of the
enqueue loop with the device time of the same K steps.  usage: python tools/host_enqueue_probe.py [workload] [K]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from uresnet_amd import uresnet  # noqa: E402
from uresnet_amd import synthetic_io as sio  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "cfg1_2d256_f16_b4"
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    dims, F, ncls, B = bench.WORKLOADS[wl][:4]
    prec = "bf16" if wl.endswith("bf16") else "fp32"
    net = uresnet(dims=list(dims), num_class=ncls, base_num_outputs=F)
    net.construct(trainable=True, use_weight=True, learning_rate=1e-3, seed=1, precision=prec)
    b = [sio.lartpc_sparse(dims, ncls, i) for i in range(B)]
    data, label, weight = (np.stack([x[j] for x in b]) for j in range(3))
    d, l, w = (torch.from_numpy(x).cuda() for x in (data, label, weight))

    def step():
        net.zero_gradients(None)
        net.accum_gradients(None, d, l, w, fetch=False)
        net.apply_gradients(None)
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%s: host enqueue %.3f ms/step, device complete %.3f ms/step (K = %d)" % (wl, (t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3, K))


if __name__ == "__main__":
    main()
