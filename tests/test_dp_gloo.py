"""world_size-2 data parallelism on CPU (gloo): the N>1 path of the product is
`ssnet_base.allreduce_gradients` (SUM over ranks of the flat gradient buffer) followed by an identical Adam
update on every rank, which must equal the reference's sequential accumulation over NUM_MINIBATCHES = world
(lib/ssnet.py:77, lib/ssnet_trainval.py:164-191).  Gradients come from the oracle here (no GPU)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import uresnet_amd  # noqa: F401
    from uresnet_amd import ssnet_base
    from uresnet_amd.ssnet_trainval import ssnet_trainval
    from oracle import uresnet_np as O
    from _net import make_inputs
    dims, base, ncls, ns = (16, 16, 1), 4, 3, 2
    P = O.init_params(2, 1, base, ncls, seed=3, num_strides=ns)
    data, label, weight = make_inputs(dims, ncls, 2, seed=50 + rank)      # every rank its own minibatch
    g, m = O.step_gradients(P, dims, base, data, label, weight, num_strides=ns)
    flat = torch.from_numpy(np.concatenate([g[k].ravel() for k in P]))

    class Fake(object):
        pass
    fake = Fake()
    fake._grads = flat.clone()
    ssnet_base.allreduce_gradients(fake)                                    # the product's reduction
    drv = Fake()
    drv._net = Fake()
    drv._net._device = "cpu"
    mets = ssnet_trainval._mean_over_ranks(drv, np.array([m["loss"], m["acc_all"], m["acc_nonzero"]]))
    if rank == 0:
        q.put((fake._grads.numpy(), mets, flat.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_dp_equals_two_minibatch_accumulation():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import uresnet_np as O
    from _net import make_inputs
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    summed, mets, own = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    dims, base, ncls, ns = (16, 16, 1), 4, 3, 2
    P = O.init_params(2, 1, base, ncls, seed=3, num_strides=ns)
    mbs = [make_inputs(dims, ncls, 2, seed=50 + r) for r in range(2)]
    P2 = type(P)((k, v.copy()) for k, v in P.items())
    ref_mets, acc = O.train_step(P2, O.Adam(P2), dims, base, mbs, num_strides=ns)
    ref = np.concatenate([acc[k].ravel() for k in P])
    assert np.allclose(summed, ref, rtol=1e-12, atol=1e-15)            # SUM, not mean
    assert not np.allclose(own, ref)
    assert np.allclose(mets, ref_mets)                                  # reported metrics: mean over ranks (:211)
