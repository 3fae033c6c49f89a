"""Consistency check at realistic sizes: the specialised kernels (tiled / igemm / pointwise / scatter-tiled, second
stream) against the generic gather kernels (URSN_DISABLE_TILED=1, URSN_WGRAD_STREAM=0) on the same weights/batch.
Run on a GPU box:  python tools/ab_paths.py 3d|2d
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = {"3d": ((96, 96, 96, 1), 8, 3, 2), "2d": ((512, 512, 1), 16, 5, 3), "3db": ((64, 96, 160, 1), 8, 3, 3),
         "3dbig": ((256, 256, 256, 1), 8, 3, 4)}   # tensors above 2^31 bytes


def child(which, tag):
    sys.path.insert(0, ROOT)
    import numpy as np
    import uresnet_amd  # noqa: F401
    from uresnet_amd import uresnet
    from uresnet_amd import synthetic_io as sio
    dims, base, ncls, N = CASES[which]
    net = uresnet(dims=list(dims), num_class=ncls, base_num_outputs=base)
    net.construct(trainable=True, use_weight=True, learning_rate=1e-3, seed=99)
    data = np.stack([sio.lartpc_sparse(dims, ncls, i)[0] for i in range(N)])
    label = np.stack([sio.lartpc_sparse(dims, ncls, i)[1] for i in range(N)])
    weight = np.stack([sio.lartpc_sparse(dims, ncls, i)[2] for i in range(N)])
    weight /= weight.sum(axis=1, keepdims=True)
    net.zero_gradients(None)
    res, _ = net.accum_gradients(None, data, label, weight)
    g = net.get_gradients()
    net.apply_gradients(None)
    res2, _ = net.accum_gradients(None, data, label, weight)
    np.savez("/tmp/ab_%s_%s.npz" % (which, tag), loss=res[1], loss2=res2[1], acc=res[2], **{k.replace("/", "|"): v for k, v in g.items()})
    print(tag, "loss", res[1], "acc", res[2], res[3], "loss after 1 step", res2[1])


def main():
    which = sys.argv[1]
    if len(sys.argv) > 2:
        return child(which, sys.argv[2])
    import numpy as np
    for tag, env in [("generic", {"URSN_DISABLE_TILED": "1", "URSN_WGRAD_STREAM": "0"}), ("fast", {})]:
        subprocess.check_call([sys.executable, os.path.abspath(__file__), which, tag], env=dict(os.environ, **env))
    a, b = np.load("/tmp/ab_%s_generic.npz" % which), np.load("/tmp/ab_%s_fast.npz" % which)
    worst = []
    for k in a.files:
        if k in ("loss", "loss2", "acc"):
            continue
        na = np.linalg.norm(a[k])
        if na > 0:
            worst.append((float(np.linalg.norm(a[k] - b[k]) / na), k.replace("|", "/")))
    worst.sort(reverse=True)
    out = {"case": which, "loss_generic": float(a["loss"]), "loss_fast": float(b["loss"]),
           "loss2_generic": float(a["loss2"]), "loss2_fast": float(b["loss2"]), "worst_grad_l2": worst[:5]}
    print(json.dumps(out, indent=1))
    assert abs(out["loss_generic"] - out["loss_fast"]) < 1e-5 * abs(out["loss_generic"])
    assert worst[0][0] < 2e-2, worst[0]  # deep-level gradients carry 0.1-1 % fp32 conditioning noise


if __name__ == "__main__":
    main()
