"""Generates the golden fixtures in this directory from the CPU oracle (oracle/uresnet_np.py, fp64),
cross-checked against the independent torch-CPU formulation before writing.

PARITY UNPINNED: the reference holds no golden vectors and its TensorFlow arithmetic cannot be run here,
so these vectors pin the *restatement* (SURVEY.md Appendix A/B semantics); they make regressions of the
oracle and of the HIP path visible and travel to the GPU box, where /root/reference does not exist.

    python tests/golden/make_golden.py          # rewrites *.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import uresnet_np as O  # noqa: E402
from oracle import uresnet_torch as T  # noqa: E402

NETS = {
    # name: dims, base, classes, batch, num_strides, use_weight
    "net2d_32x32_f4_ns3": ((32, 32, 1), 4, 3, 2, 3, False),
    "net3d_16x16x16_f4_ns2": ((16, 16, 16, 1), 4, 3, 2, 2, True),
}


def make_net(name, dims, base, ncls, N, ns, use_w):
    import torch
    from _net import make_inputs
    nd = len(dims) - 1
    P = O.init_params(nd, dims[-1], base, ncls, seed=2024, beta_scale=0.2, num_strides=ns)
    P = type(P)((k, v.astype(np.float32).astype(np.float64)) for k, v in P.items())
    data, label, weight = make_inputs(dims, ncls, N, seed=77)
    w = weight if use_w else None
    g, m = O.step_gradients(P, dims, base, data, label, w, num_strides=ns)
    Pt = T.params_from_numpy(P)
    gt, mt = T.step_gradients(Pt, dims, base, data, label, w, num_strides=ns)
    assert abs(m["loss"] - mt["loss"]) < 1e-10 * abs(m["loss"])
    for k in g:
        assert np.abs(g[k] - gt[k].numpy()).max() <= 1e-9 * (np.abs(g[k]).max() + 1e-30), k
    # two Adam iterations of NUM_MINIBATCHES=1 on the same batch
    P2 = type(P)((k, v.copy()) for k, v in P.items())
    opt = O.Adam(P2, lr=1e-3)
    losses = []
    for _ in range(2):
        mets, _ = O.train_step(P2, opt, dims, base, [(data, label, weight)], use_weight=use_w, num_strides=ns)
        losses.append(mets[0])
    out = dict(dims=np.array(dims), base=base, num_class=ncls, num_strides=ns, use_weight=int(use_w),
               data=data, label=label, weight=weight,
               loss=m["loss"], acc_all=m["acc_all"], acc_nonzero=m["acc_nonzero"],
               logits=m["logits"].astype(np.float32), softmax=m["softmax"].astype(np.float32),
               adam_losses=np.array(losses))
    for k, v in P.items():
        out["param:" + k] = v.astype(np.float32)
        out["grad:" + k] = g[k].astype(np.float32)
        out["adam2:" + k] = P2[k].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "loss", m["loss"], "params", sum(v.size for v in P.values()))


def make_ops():
    rng = np.random.default_rng(99)
    out = {}
    for tag, nd, S, ci, co, k, s in [("c3s1", 3, (5, 6, 7), 3, 4, 3, 1), ("c3s2", 3, (4, 6, 8), 3, 5, 3, 2),
                                     ("c1s2", 2, (6, 8), 4, 6, 1, 2), ("c2s2odd", 2, (5, 7), 2, 3, 3, 2)]:
        x = rng.standard_normal((2,) + S + (ci,))
        w = rng.standard_normal((k,) * nd + (ci, co))
        y = O.conv_fwd(x, w, s)
        dy = rng.standard_normal(y.shape)
        dx, dw = O.conv_bwd(x, w, s, dy)
        out.update({tag + ":x": x, tag + ":w": w, tag + ":y": y, tag + ":dy": dy, tag + ":dx": dx, tag + ":dw": dw,
                    tag + ":stride": s})
    for tag, nd, S, ci, co in [("d3", 3, (3, 4, 5), 4, 3), ("d2", 2, (4, 6), 5, 2)]:
        x = rng.standard_normal((2,) + S + (ci,))
        w = rng.standard_normal((3,) * nd + (co, ci))
        y = O.deconv_fwd(x, w)
        dy = rng.standard_normal(y.shape)
        dx, dw = O.deconv_bwd(x, w, dy)
        out.update({tag + ":x": x, tag + ":w": w, tag + ":y": y, tag + ":dy": dy, tag + ":dx": dx, tag + ":dw": dw})
    np.savez_compressed(os.path.join(HERE, "ops.npz"), **out)
    print("ops", len(out))


if __name__ == "__main__":
    for name, cfg in NETS.items():
        make_net(name, *cfg)
    make_ops()
