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


# ---- the DRIVER under a 2-rank group: entry sharding, rank-0-only report / log / snapshot, metric
# averaging as a collective only on report iterations, one all-reduce + Adam per iteration ------------------------------
class _StubNet(object):
    """Stands in for ``uresnet`` on a machine without a GPU: records the call sequence the driver issues and returns
    metrics derived from the entries it was fed (so the per-rank values differ and the mean is checkable)."""
    log = []

    def __init__(self, dims, num_class, base_num_outputs=16, debug=False):
        self._dims, self._num_class = list(dims), num_class
        self._device = "cpu"

    def construct(self, trainable=True, use_weight=True, learning_rate=None, seed=1234, precision="fp32"):
        import uresnet_amd.ssnet as S
        self._opt = S._Adam(learning_rate if learning_rate and learning_rate > 0 else 1e-3)
        self._trainable = trainable
        self._grads = torch.zeros(8, dtype=torch.float64)
        self._vars = {"UResNet/conv0/weights": np.zeros(3, np.float32)}

    def zero_gradients(self, sess=None):
        self._grads.zero_()
        self.log.append("zero")

    def accum_gradients(self, sess, input_data, input_label, input_weight=None, fetch=True):
        assert abs(float(input_weight.sum(axis=1)[0]) - 1.0) < 1e-5          # normalised in place by the driver (:173)
        first = float(input_data[0, 0])                                       # dense_uniform: identifies the entry
        self._grads += 1.0
        self._last_feed = {"input_data": input_data, "input_label": input_label, "input_weight": input_weight}
        self.log.append(("accum", fetch, first))
        doc = ['', 'loss', 'acc. all', 'acc. nonzero']
        return ([None, 10.0 * (dist.get_rank() + 1), 0.5, 0.25] if fetch else None), doc

    def last_feed(self):
        return dict(self._last_feed)

    def apply_gradients(self, sess=None):
        from uresnet_amd import ssnet_base
        ssnet_base.allreduce_gradients(self)
        self.log.append(("apply", float(self._grads[0])))

    def run_test(self, sess, d, l, w=None):
        return [1.0, 1.0, 1.0], ['loss', 'acc. all', 'acc. nonzero']

    def make_summary(self, sess, d, l, w=None):
        return {'loss': 1.0, 'accuracy_all': 1.0, 'accuracy_nonzero': 1.0}

    def variable_names(self):
        return list(self._vars)

    def get_variables(self):
        return dict(self._vars)


def _driver_worker(rank, world, port, tmp, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import io
    from contextlib import redirect_stdout
    import uresnet_amd  # noqa: F401
    from uresnet_amd import synthetic_io as sio
    tv = sys.modules["uresnet_amd.ssnet_trainval"] if "uresnet_amd.ssnet_trainval" in sys.modules else None
    if tv is None:
        import importlib
        tv = importlib.import_module("uresnet_amd.ssnet_trainval")
    tv.uresnet = _StubNet
    tv.HipSession = lambda: None
    tv.ssnet_trainval.report_memory = lambda self: 0.0
    inp = os.path.join(tmp, "input.cfg")
    cfg = os.path.join(tmp, "train.cfg")
    if rank == 0:
        with open(inp, "w") as f:
            f.write("Dims [16, 16, 1]\nNumClass 3\nGenerator 'dense_uniform'\nNumEntries 1000\n"
                    "Keys {'data': 'data', 'label': 'label', 'weight': 'weight'}\n")
        with open(cfg, "w") as f:
            f.write("NUM_CLASS 3\nBASE_NUM_FILTERS 4\nMAIN_INPUT_CONFIG '%s'\nLOGDIR '%s'\nSAVE_FILE '%s'\n"
                    "ITERATIONS 4\nMINIBATCH_SIZE 2\nNUM_MINIBATCHES 2\nLEARNING_RATE 0.001\nTRAIN True\n"
                    "USE_WEIGHTS True\nREPORT_STEPS 2\nSUMMARY_STEPS 2\nCHECKPOINT_STEPS 2\n"
                    % (inp, os.path.join(tmp, "log"), os.path.join(tmp, "ckpt", "net")))
    dist.barrier()
    out = io.StringIO()
    with redirect_stdout(out):
        t = tv.ssnet_trainval()
        t.override_config(cfg)
        t.initialize()
        t.batch_process()
        t.reset()
    first_vals = [sio.dense_uniform([16, 16, 1], 3, e)[0][0] for e in range(32)]
    fed = [first_vals.index(np.float32(x[2])) for x in _StubNet.log if isinstance(x, tuple) and x[0] == "accum"]
    q.put((rank, out.getvalue(), fed, [x for x in _StubNet.log if not (isinstance(x, tuple) and x[0] == "accum")],
           [x[1] for x in _StubNet.log if isinstance(x, tuple) and x[0] == "accum"]))
    dist.barrier()
    dist.destroy_process_group()


def test_driver_under_two_ranks(tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_driver_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(2):
        r = q.get(timeout=300)
        got[r[0]] = r[1:]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # entry sharding: rank r reads entries r, r+2, r+4, ... two per minibatch, two minibatches per iteration
    assert got[0][1] == [0, 4, 8, 12, 16, 20, 24, 28] and got[1][1] == [1, 5, 9, 13, 17, 21, 25, 29]
    # one zero + one all-reduce/Adam per iteration; the reduction is a SUM over ranks of 2 minibatches each
    assert got[0][2] == ["zero", ("apply", 4.0)] * 4 == got[1][2]
    # the three scalars are fetched (a stream synchronisation) only on report iterations (0 and 2)
    assert got[0][3] == [True, True, False, False, True, True, False, False]
    # only rank 0 prints / logs / saves; the reported train metric is the mean over ranks of (10, 20)
    assert got[1][0].count("@ iteration") == 0 and "saved @" not in got[1][0]
    assert got[0][0].count("@ iteration") == 2 and got[0][0].count("saved @") == 2
    assert "Train set: loss=15.000000   acc. all=0.500000   acc. nonzero=0.250000" in got[0][0]
    assert sorted(os.listdir(str(tmp_path / "ckpt"))) == ["net-1.npz", "net-3.npz"]
    lines = (tmp_path / "log" / "train" / "scalars.jsonl").read_text().strip().split("\n")
    assert [__import__("json").loads(x)["iteration"] for x in lines] == [0, 2]
