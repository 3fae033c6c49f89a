"""Two data-parallel ranks on ONE MI355X (both processes on cuda:0, gloo process group -- RCCL refuses two ranks on one
device): the product's own N > 1 path end to end.  Every rank builds the net from the same seed, takes its own minibatch through
zero_gradients -> accum_gradients (HIP kernels) -> apply_gradients (all_reduce(SUM) of the flat gradient buffer, then Adam),
and must end with weights that are bitwise equal on both ranks and bitwise equal to ONE process that accumulates the two
minibatches (NUM_MINIBATCHES = 2: lib/ssnet.py:77 sums gradients, lib/ssnet_trainval.py:164-191) before its update -- the sum of
two terms does not depend on who adds them.  The 8-GPU RCCL run itself is the driver's (bench.py --gpus N)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_RANK = r"""
import os, sys
import numpy as np
root, out, mode = sys.argv[1], sys.argv[2], sys.argv[3]
sys.path.insert(0, root); sys.path.insert(0, root + "/tests")
import torch
import torch.distributed as dist
from _net import make_inputs
from uresnet_amd import uresnet
rank = int(os.environ.get("RANK", "0"))
if mode == "dp":
    dist.init_process_group("gloo", rank=rank, world_size=2)
dims, base, ncls, ns = (32, 32, 32, 1), 8, 3, 2
net = uresnet(dims=list(dims), num_class=ncls, base_num_outputs=base, num_strides=ns)
net.construct(trainable=True, use_weight=True, learning_rate=1e-2, seed=7)
batches = [make_inputs(dims, ncls, 2, seed=60 + r) for r in range(2)]
for it in range(2):
    net.zero_gradients(None)
    if mode == "dp":
        net.accum_gradients(None, *batches[rank])
    else:
        for b in batches:
            net.accum_gradients(None, *b)
    net.apply_gradients(None)
v = net.get_variables()
np.savez(out, **{k.replace("/", "|"): a for k, a in v.items()})
if mode == "dp":
    dist.barrier()
    dist.destroy_process_group()
"""


def test_two_ranks_on_one_gpu_equal_two_minibatch_accumulation(tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 300), WORLD_SIZE="2",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    outs = [str(tmp_path / ("rank%d.npz" % r)) for r in range(2)]
    procs = [subprocess.Popen([sys.executable, "-c", _RANK, ROOT, outs[r], "dp"], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    for p in procs:
        so, se = p.communicate(timeout=600)
        assert p.returncode == 0, se[-2000:]
    single = str(tmp_path / "single.npz")
    p = subprocess.run([sys.executable, "-c", _RANK, ROOT, single, "single"], env=os.environ.copy(), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    a, b, s = np.load(outs[0]), np.load(outs[1]), np.load(single)
    assert set(a.files) == set(s.files) and len(a.files) >= 40
    moved = 0
    for k in a.files:
        assert np.array_equal(a[k], b[k]), ("ranks diverged", k)
        assert np.array_equal(a[k], s[k]), ("data parallel != sequential accumulation", k)
        moved += int(np.abs(a[k]).sum() > 0)
    assert moved >= 40


def test_bench_n2_path_two_ranks_on_one_gpu_over_gloo():
    """bench.py's world > 1 path (barrier + max-over-ranks timing, all-reduce inside apply_gradients, rank-0 JSON line) end to
    end as the driver launches it (`python bench.py --gpus 2` -> torch.distributed.run, one process per rank), here with both
    ranks on the one GPU over gloo (--backend gloo; RCCL refuses two ranks on one device).  The 8-GPU RCCL run is the driver's."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--workload", "tiny_3d64_f8_b2", "--secondary-workload", "tiny_3d64_f8_b2_bf16",
                        "--backend", "gloo"], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [x for x in p.stdout.split("\n") if x.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]           # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["steps"] == 2 and out["warmup"] == 1
    assert out["config"]["global_batch"] == 4 and out["config"]["parallelism"] == "dp2" and out["config"]["backend"] == "gloo"
    assert out["scaling"] == "weak" and out["value"] > 0 and abs(out["value"] - 4 * 2 / (out["ms_per_step"] * 2e-3)) < 1e-2 * out["value"]
    assert out["step_parts"]["allreduce_ms"] > 0
    assert out["roofline"] is not None and out["cpu_baseline"] is None
    # the line is self-checking for the day 8 GPUs are there: every rank's own clock over the timed region and the device it bound
    pr = out["per_rank"]
    assert len(pr["ms_per_step"]) == 2 and pr["device_index"] == [0, 0]          # gloo rehearsal: both ranks share device 0
    assert pr["ms_per_step_min"] <= pr["ms_per_step_max"] and abs(pr["ms_per_step_max"] - out["ms_per_step"]) < 1e-2 * out["ms_per_step"]
    # the secondary (bf16) leg rides the same N: a child job of two ranks started after this one released its GPUs
    sec = out["secondary"]
    assert "error" not in sec, sec
    assert sec["workload"] == "tiny_3d64_f8_b2_bf16" and sec["dtype"] == "bf16" and sec["n_gpus"] == 2 and sec["ranks_seen"] == 2
    assert sec["config"]["global_batch"] == 4 and sec["value"] > 0 and sec["step_parts"]["allreduce_ms"] > 0
    assert len(sec["per_rank"]["ms_per_step"]) == 2
