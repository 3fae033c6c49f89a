#!/usr/bin/env python
"""bench.py -- fwd+bwd images/s of the MI355X-native U-ResNet hot path (BASELINE.json metric).

A "step" is one full reference training iteration with NUM_MINIBATCHES=1 on device-resident synthetic
LArTPC volumes: zero_gradients -> accum_gradients (forward + loss + backward) -> [RCCL sum-all-reduce
of the flat gradient buffer when N > 1] -> apply_gradients (TF-form Adam)
(lib/ssnet_trainval.py:164-191).  Default workload: BASELINE.json configs[2], the 3-D 192^3x1 3-class
F=8 U-ResNet at batch 4 per GPU, fp32.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     -- the dominant kernel: algorithmic FLOPs / HIP-event launch durations vs the fp32 peak
  cpu_baseline -- the torch-CPU restatement of the reference graph (oracle "port") timed on the host
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 vector == fp32 MFMA peak
PEAK_HBM_GBS = 8000.0      # spec; ~6300 achievable
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA

WORKLOADS = {
    # name: dims, base filters, classes, per-GPU batch, generator
    "cfg3_3d192_f8_b4": ((192, 192, 192, 1), 8, 3, 4, "lartpc_sparse"),
    "cfg2_2d512_f16_b16": ((512, 512, 1), 16, 5, 16, "lartpc_sparse"),
    "cfg1_2d256_f16_b4": ((256, 256, 1), 16, 3, 4, "dense_uniform"),
    "tiny_3d64_f8_b2": ((64, 64, 64, 1), 8, 3, 2, "lartpc_sparse"),
    # BASELINE.json configs[4]: 3-D 256^3 bf16 mixed precision (batch 4 per GPU) -- graded against HBM
    "cfg5_3d256_f8_b4_bf16": ((256, 256, 256, 1), 8, 3, 4, "lartpc_sparse"),
    # the same volume in fp32: 96.5 GB of workspace, sized for 288 GB HBM
    "cfg5shape_3d256_f8_b4_fp32": ((256, 256, 256, 1), 8, 3, 4, "lartpc_sparse"),
    "tiny_3d64_f8_b2_bf16": ((64, 64, 64, 1), 8, 3, 2, "lartpc_sparse"),
}


def algorithmic_work(dims, F, ncls, batch, elem):
    """SURVEY.md 8(d) conventions: FLOPs = 2 MACs of the 58 conv-like layers, bwd = dgrad + wgrad (conv0 has no dgrad);
    bytes per pass = x + y + w with activations at `elem` bytes and weights / weight gradients fp32.  Returns
    (flops fwd+bwd, bytes fwd+bwd, T_roof seconds against `peak_flops`, 8 TB/s) through a closure."""
    nd = len(dims) - 1
    vox0 = 1
    for d in dims[:-1]:
        vox0 *= int(d)
    layers = []   # (k, cin, cout, in level, out level, has_dgrad)

    def unit(ci, co, s, lin, lout):
        if not (ci == co and s == 1):
            layers.append((1, ci, co, lin, lout, True))
        layers.append((3, ci, co, lin, lout, True))
        layers.append((3, co, co, lout, lout, True))
    layers.append((3, int(dims[-1]), F, 0, 0, False))
    c = F
    for step in range(5):
        unit(c, 2 * c, 2, step, step + 1)
        unit(2 * c, 2 * c, 1, step + 1, step + 1)
        c *= 2
    for step in range(5):
        lvl = 4 - step
        layers.append((3, c, c // 2, lvl + 1, lvl, True))       # deconv: MACs counted on the INPUT grid
        unit(c, c // 2, 1, lvl, lvl)
        unit(c // 2, c // 2, 1, lvl, lvl)
        c //= 2
    layers.append((3, c, F, 0, 0, True))
    layers.append((3, F, ncls, 0, 0, True))
    out = []
    for i, (k, ci, co, lin, lout, dg) in enumerate(layers):
        vin, vout = batch * vox0 / (2 ** nd) ** lin, batch * vox0 / (2 ** nd) ** lout
        is_deconv = lin > lout
        macs = (vin if is_deconv else vout) * (k ** nd) * ci * co
        byts = elem * (vin * ci + vout * co) + 4.0 * (k ** nd) * ci * co
        out.append((2.0 * macs, byts, dg))
    return out


def step_roofline(dims, F, ncls, batch, elem, peak_flops, peak_bw):
    fl = by = t = 0.0
    for f, b, dg in algorithmic_work(dims, F, ncls, batch, elem):
        passes = 3 if dg else 2
        fl += passes * f
        by += passes * b
        t += max(f / peak_flops, b / peak_bw) + max((passes - 1) * f / peak_flops, (passes - 1) * b / peak_bw)
    return fl, by, t


def kernel_family(name):
    """'b3conv_bf16<8,8>+bn' -> 'b3conv', 'b3wgrad_bf16<8,8>(pair)' -> 'b3wgrad', 'twgradz_kernel<3>' -> 'twgradz'."""
    base = name.split("<")[0].split("(")[0].split("+")[0].strip()
    for suf in ("_bf16", "_kernel"):
        if base.endswith(suf):
            base = base[:-len(suf)]
    return base


def group_families(by_kernel):
    """Per-launch HIP-event records grouped by kernel family; roof_ms = sum over launches of max(F / P, B / BW)."""
    fam = {}
    for k, v in by_kernel.items():
        f = kernel_family(k)
        if not f or v["ms"] <= 0:
            continue
        e = fam.setdefault(f, dict(ms=0.0, flops=0.0, bytes=0.0, launches=0, roof_ms=0.0, conv=v["conv"], members=[]))
        e["ms"] += v["ms"]; e["flops"] += v["flops"]; e["bytes"] += v["bytes"]; e["launches"] += v["launches"]
        e["roof_ms"] += v.get("roof_ms", 0.0)
        e["members"].append(k)
    return fam


def cpu_baseline_child(kind, size, reps, warm):
    """Runs in a child process: the torch-CPU oracle, fwd + loss + bwd (one accum_gradients-equivalent).
    kind '3d': ONE image of size^3 x1 with cfg3's model (F=8, 3 classes, lartpc_sparse);
    kind 'cfg1': BASELINE configs[0] exactly (2-D 256^2, F=16, 3 classes, batch 4, dense_uniform, USE_WEIGHTS False)."""
    import numpy as np
    import torch
    from oracle import uresnet_np as O, uresnet_torch as T
    from importlib import import_module
    import uresnet_amd  # noqa: F401
    sio = import_module("uresnet_amd.synthetic_io")
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    torch.set_num_threads(max(1, min(ncpu, 16)))   # one-GPU box share is 16 cores (the host reports 256)
    if kind == "cfg1":
        dims, base, ncls, n, gen, use_w = (256, 256, 1), 16, 3, 4, sio.dense_uniform, False
    else:
        dims, base, ncls, n, gen, use_w = (size, size, size, 1), 8, 3, 1, sio.lartpc_sparse, True
    P = T.params_from_numpy(O.init_params(len(dims) - 1, 1, base, ncls, seed=1234, dtype=np.float32), dtype=torch.float32)
    b = [gen(dims, ncls, i) for i in range(n)]
    d, l, w = (np.stack([x[j] for x in b]) for j in range(3))
    w = w / w.sum(axis=1, keepdims=True)
    times = []
    for i in range(warm + reps):
        t0 = time.time()
        T.step_gradients(P, dims, base, d, l, w if use_w else None)
        dt = time.time() - t0
        sys.stderr.write("cpu_baseline %s rep %d%s: %.2f s on %d threads\n" % (kind, i, " (warm-up)" if i < warm else "", dt,
                                                                             torch.get_num_threads()))
        sys.stderr.flush()
        if i >= warm:
            times.append(dt)
    times.sort()
    print(json.dumps({"sec_per_step": times[len(times) // 2], "images": n, "threads": torch.get_num_threads(),
                      "size": size, "reps": reps, "warm": warm}))


def _cpu_child(kind, size, reps, warm, mkldnn, timeout):
    env = dict(os.environ, URSN_ORACLE_MKLDNN=mkldnn)
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", kind, "--cpu-size", str(size),
                              "--cpu-reps", str(reps), "--cpu-warm", str(warm)], env=env, stdout=subprocess.PIPE,
                             text=True, timeout=timeout)
        line = [x for x in out.stdout.strip().split("\n") if x.startswith("{")]
        if out.returncode != 0 or not line:
            return None
        return json.loads(line[-1])
    except Exception:
        return None


def run_cpu_baseline(full_size, reps=5, warm=2):
    """the torch-CPU port on ONE full 192^3 image, median of >= 5 after 2 warm-ups (oneDNN convs; if
    that build crashes here the ATen-native path at 128^3, scaled by voxel count and labelled so), plus cfg1 exactly."""
    attempts = [(full_size, "1", reps, warm, 600), (min(full_size, 128), "0", 3, 1, 600)]
    out = None
    for size, mkldnn, r, wm, to in attempts:
        res = _cpu_child("3d", size, r, wm, mkldnn, to)
        if res is None:
            continue
        scale = (float(size) / full_size) ** 3   # work is linear in voxels
        sec_full = res["sec_per_step"] / scale
        out = {"value": round(1.0 / sec_full, 5), "unit": "images/s", "cores": res["threads"], "kind": "port",
               "sample": "torch-CPU restatement of the reference graph (not TensorFlow), fp32, %s convs: ONE image of "
                         "%d^3x1 (F=8, 3 classes) fwd+loss+bwd, median of %d after %d warm-ups (%.2f s per image); %s"
                         % ("oneDNN" if mkldnn == "1" else "ATen-native", size, r, wm, res["sec_per_step"],
                            "measured at the config's full size" if size == full_size else
                            "FALLBACK: oneDNN crashed at full size, scaled by voxel count to %d^3" % full_size)}
        break
    c1 = _cpu_child("cfg1", 256, reps, warm, "0", 300)   # oneDNN's 2-D backward segfaults intermittently in this image
    if out is not None and c1 is not None:
        out["cfg1_exact"] = {"value": round(c1["images"] / c1["sec_per_step"], 3), "unit": "images/s",
                             "sample": "BASELINE configs[0] exactly: 2-D 256^2x1, F=16, 3 classes, batch 4, dense_uniform, "
                                       "USE_WEIGHTS False; ATen-native convs, median of %d after %d warm-ups" % (reps, warm)}
    return out


def self_launch(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N-rank job as a CHILD process (nothing in this
    process has touched the GPU yet -- never re-exec a process that has) and relay its JSON line and exit code."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.run(cmd, env=env)
    sys.exit(p.returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg3_3d192_f8_b4", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch override")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="per-kernel time table on stderr")
    ap.add_argument("--layers", action="store_true", help="per-(layer, pass) time table on stderr")
    ap.add_argument("--host-feed", action="store_true",
                    help="feed host (numpy) batches through the pinned copy-stream path every step; value stays the "
                         "device-resident rate, the PCIe-inclusive rate is reported beside it")
    ap.add_argument("--backend", default=os.environ.get("URSN_DIST_BACKEND", "nccl"), choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1: nccl (= RCCL over xGMI, the production path) or gloo "
                         "(lets several ranks share ONE GPU: the N > 1 code path rehearsed on a one-GPU box)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary workload (cfg5 bf16) that the default cfg3 run appends to its JSON line")
    ap.add_argument("--secondary-workload", default="", choices=[""] + sorted(WORKLOADS),
                    help="workload of the secondary leg (default: cfg5_3d256_f8_b4_bf16 behind the default cfg3 run; naming one "
                         "attaches it to any primary workload -- the N > 1 rehearsal uses the tiny pair)")
    ap.add_argument("--cpu-baseline-child", default="")
    ap.add_argument("--cpu-size", type=int, default=192)
    ap.add_argument("--cpu-reps", type=int, default=5)
    ap.add_argument("--cpu-warm", type=int, default=2)
    args = ap.parse_args()
    if args.cpu_baseline_child:
        return cpu_baseline_child(args.cpu_baseline_child, args.cpu_size, args.cpu_reps, args.cpu_warm)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)

    import numpy as np
    import torch
    import torch.distributed as dist
    import uresnet_amd  # noqa: F401
    from uresnet_amd import uresnet, _lib
    from uresnet_amd import synthetic_io as sio
    import ctypes

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback for the product path)"
    ndev = torch.cuda.device_count()
    dev_index = local_rank if args.backend == "nccl" else local_rank % max(ndev, 1)   # gloo: ranks may share a device
    torch.cuda.set_device(dev_index)
    ranks_seen = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=args.backend)   # "nccl" IS RCCL over xGMI on ROCm
        ranks_seen = dist.get_world_size()
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d; launch with --nproc-per-node == --gpus\n" % (args.gpus, world))
        sys.exit(2)

    dims, base, ncls, batch, gen = WORKLOADS[args.workload]
    if args.batch:
        batch = args.batch
    bf16 = args.workload.endswith("_bf16")
    net = uresnet(dims=list(dims), num_class=ncls, base_num_outputs=base)
    net.construct(trainable=True, use_weight=True, learning_rate=1e-4, seed=1234, precision="bf16" if bf16 else "fp32")

    # device-resident synthetic batch, per rank its own entries (weak scaling: per-GPU work fixed)
    g = sio.GENERATORS[gen]
    dsz, lsz = int(np.prod(dims)), int(np.prod(dims[:-1]))
    data = np.empty((batch, dsz), np.float32)
    label = np.empty((batch, lsz), np.float32)
    weight = np.empty((batch, lsz), np.float32)
    for i in range(batch):
        d, l, w = g(dims, ncls, rank * batch + i)
        data[i], label[i], weight[i] = d, l, w / w.sum()   # lib/ssnet_trainval.py:173
    dev = torch.device("cuda", dev_index)
    data_d, label_d, weight_d = (torch.from_numpy(a).to(dev) for a in (data, label, weight))

    def step():
        net.zero_gradients(None)
        net.accum_gradients(None, data_d, label_d, weight_d, fetch=False)
        net.apply_gradients(None)   # all-reduce(sum) over ranks happens inside when world > 1

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    lib = _lib.load()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    # Roofline leg: per-kernel durations with HIP events on the launch stream.  The timed region above runs the
    # weight gradients on a second stream, where concurrent kernels time-slice and an event interval includes the
    # partner's work; so the same steps are repeated with that overlap switched off and every launch bracketed.
    prof_steps = min(args.steps, 3)
    _lib.check(lib.ursn_set_wgrad_overlap(net._handle, 0))
    _lib.check(lib.ursn_profile_enable(net._handle, 1))
    torch.cuda.synchronize()
    tp0 = time.perf_counter()
    for _ in range(prof_steps):
        step()
    torch.cuda.synchronize()
    serial_ms_per_step = (time.perf_counter() - tp0) / prof_steps * 1e3
    _lib.check(lib.ursn_set_wgrad_overlap(net._handle, 1))
    per_rank = None
    if world > 1:
        # every rank's own clock over the SAME barrier-bracketed region and the device it bound: a straggler GPU or two ranks
        # on one device show up in the line (the headline uses the maximum, as the contract asks)
        mine = torch.tensor([elapsed, float(dev_index)], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        el = [float(a[0].item()) for a in allr]
        per_rank = {"ms_per_step_min": round(min(el) / args.steps * 1e3, 3), "ms_per_step_max": round(max(el) / args.steps * 1e3, 3),
                    "ms_per_step": [round(e / args.steps * 1e3, 3) for e in el], "device_index": [int(a[1].item()) for a in allr]}
        elapsed = max(el)
    metrics = net.read_metrics()

    # the non-conv parts of the step, each timed alone (Adam and the all-reduce reported separately)
    def timed_ms(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        tt = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - tt) / reps * 1e3
    parts = {"zero_gradients_ms": round(timed_ms(lambda: net.zero_gradients(None)), 4),
             "allreduce_ms": round(timed_ms(net.allreduce_gradients), 4) if world > 1 else 0.0}
    # Adam last (it moves the weights; the bench is over): gradient buffer zeroed so the update is a pure decay of m, v
    net.zero_gradients(None)
    parts["adam_ms"] = round(timed_ms(lambda: _lib.check(lib.ursn_apply_adam(net._handle, 1e-4, None))), 4)

    # per-launch HIP-event records of the timed region (rank 0)
    cnt = ctypes.c_int64(0)
    _lib.check(lib.ursn_profile_read(net._handle, None, 0, ctypes.byref(cnt)))
    nrec = 1 << 20
    recs = (_lib.ursn_prof_rec * nrec)()
    _lib.check(lib.ursn_profile_read(net._handle, recs, nrec, ctypes.byref(cnt)))
    _lib.check(lib.ursn_profile_enable(net._handle, 0))
    host_feed = None
    if args.host_feed:
        # the same step fed from host memory every iteration (lib/ssnet_trainval.py:167-188 hands over host buffers):
        # pinned source, copy stream, copy of step k+1 overlapped with the kernels of step k (ssnet.py::_feed)
        pin = [torch.from_numpy(a).pin_memory().numpy() for a in (data, label, weight)]

        def hstep():
            net.zero_gradients(None)
            net.accum_gradients(None, pin[0], pin[1], pin[2], fetch=False)
            net.apply_gradients(None)
        for _ in range(2):
            hstep()
        barrier()
        th = time.perf_counter()
        for _ in range(args.steps):
            hstep()
        barrier()
        h_el = time.perf_counter() - th
        nbytes = sum(a.nbytes for a in pin)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cs = net._copy_stream()
        with torch.cuda.stream(cs):
            ev0.record(cs)
            for a, dd in zip(pin, (data_d, label_d, weight_d)):
                dd.copy_(torch.from_numpy(a), non_blocking=True)
            ev1.record(cs)
        ev1.synchronize()
        host_feed = {"images_per_s_pcie_inclusive": round(batch * world * args.steps / h_el, 3),
                     "ms_per_step": round(h_el / args.steps * 1e3, 3), "h2d_bytes_per_step": nbytes,
                     "h2d_ms_alone": round(ev0.elapsed_time(ev1), 3),
                     "note": "pinned source buffers, copy stream, H2D of step k+1 overlaps the kernels of step k"}

    by_kernel, t_roof_ms, conv_flops, conv_bytes, all_ms = {}, 0.0, 0.0, 0.0, 0.0
    peak_flops_step = PEAK_BF16_TFLOPS if bf16 else PEAK_FP32_TFLOPS   # TFLOP/s the conv kernels of this plan are priced against
    by_layer = {}
    for i in range(cnt.value):
        r = recs[i]
        k = r.kernel.decode()
        le = by_layer.setdefault((r.layer.decode(), r.pass_, k), [0.0, r.flops, r.bytes])
        le[0] += r.ms
        e = by_kernel.setdefault(k, dict(ms=0.0, flops=0.0, bytes=0.0, launches=0, conv=r.pass_ <= 2, roof_ms=0.0))
        e["ms"] += r.ms; e["flops"] += r.flops; e["bytes"] += r.bytes; e["launches"] += max(int(r.launches), 1)
        e["roof_ms"] += max(r.flops / (peak_flops_step * 1e9), r.bytes / (PEAK_HBM_GBS * 1e6))
        all_ms += r.ms
        if r.pass_ <= 2:
            t_roof_ms += max(r.flops / (PEAK_FP32_TFLOPS * 1e9), r.bytes / (PEAK_HBM_GBS * 1e6))
            conv_flops += r.flops; conv_bytes += r.bytes
    ms_per_step = elapsed / args.steps * 1e3
    value = batch * world * args.steps / elapsed

    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside the process; the per-launch figure
    # comes from the committed rocprofv3 --pmc passes of this same command (profiles/pmc_traffic.json), else null
    pmc = {}
    pmc_file = {"cfg3_3d192_f8_b4": "pmc_traffic.json", "cfg5_3d256_f8_b4_bf16": "r04_pmc_traffic_cfg5_bf16.json"}.get(args.workload)
    try:
        if pmc_file:
            with open(os.path.join(ROOT, "profiles", pmc_file)) as f:
                pmc = json.load(f)
    except Exception:
        pmc = {}
    roofline = None
    if bf16:
        # The bf16 plan is graded against HBM (30-34 of its 58 layers are bandwidth-bound at the bf16 MFMA
        # peak).  Dominant conv kernel from the per-launch HIP-event records: its algorithmic bytes (x + y + w of the layers it
        # ran) over its summed dispatch time; the whole-step figures price BatchNorm / join / head at 0 bytes, as the model does.
        fl, by, t_roof = step_roofline(dims, base, ncls, batch, 2, PEAK_BF16_TFLOPS * 1e12, PEAK_HBM_GBS * 1e9)
        ach_step = by / (ms_per_step * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": "whole step", "achieved": round(ach_step, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(ach_step / PEAK_HBM_GBS, 4), "traffic": None, "traffic_source": None}
        # kernels grouped by FAMILY (b3conv*, bconv*, b3wgrad*, bwgrad*, ...): the finer per-variant labels would otherwise let a
        # family of many small launches hide behind its largest member, or a variant stand in for the family
        fam = group_families(by_kernel)
        convs = [(k, v) for k, v in fam.items() if v["conv"]]
        if convs:
            dom_name, dom = max(convs, key=lambda kv: kv[1]["ms"])
            ach = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9
            pm = [(pmc[k]["hbm_bytes_per_launch"], pmc[k]["launches"]) for k in pmc if not k.startswith("_") and kernel_family(k) == dom_name]
            if pm:
                roofline["traffic"] = round(sum(b * l for b, l in pm) / sum(l for _, l in pm))
                roofline["traffic_source"] = ("profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, committed; "
                                              "launch-weighted over the family's kernels; not measured by this run)" % pmc_file)
            roofline.update({"kernel": dom_name + "* (family: %s)" % ", ".join(sorted(dom["members"])),
                             "achieved": round(ach, 1), "frac": round(ach / PEAK_HBM_GBS, 4),
                             "algorithmic_bytes_per_launch": round(dom["bytes"] / dom["launches"]),
                             "launches": dom["launches"], "avg_launch_ms": round(dom["ms"] / dom["launches"], 4),
                             "kernel_share_of_step": round(dom["ms"] / max(all_ms, 1e-9), 3),
                             "frac_of_own_roofline": round(dom["roof_ms"] / dom["ms"], 4),
                             "achieved_TFLOPs": round(dom["flops"] / (dom["ms"] * 1e-3) / 1e12, 1),
                             "kernel_timing": "HIP events per launch, %d steps with the weight-gradient stream serialised "
                                              "(%.1f ms/step serial vs %.1f overlapped)" % (prof_steps, serial_ms_per_step, ms_per_step)})
            # ... and the conv family furthest below its own roofline among those worth >= 2 % of the step
            low = [(k, v) for k, v in convs if v["ms"] >= 0.02 * all_ms]
            if low:
                ln, lv = min(low, key=lambda kv: kv[1]["roof_ms"] / kv[1]["ms"])
                roofline["furthest_below_roofline"] = {
                    "kernel": ln + "*", "ms_per_step": round(lv["ms"] / prof_steps, 3), "launches_per_step": lv["launches"] // prof_steps,
                    "frac_of_own_roofline": round(lv["roof_ms"] / lv["ms"], 4),
                    "achieved_GBs_algorithmic": round(lv["bytes"] / (lv["ms"] * 1e-3) / 1e9, 1),
                    "achieved_TFLOPs": round(lv["flops"] / (lv["ms"] * 1e-3) / 1e12, 1)}
            roofline["families_ms_per_step"] = {k: round(v["ms"] / prof_steps, 3) for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])}
        roofline.update({"algorithmic_bytes_per_step": round(by), "step_algorithmic_TFLOPs": round(fl / 1e12, 3),
                         "step_T_roof_ms": round(t_roof * 1e3, 3), "step_frac_of_roofline": round(t_roof * 1e3 / ms_per_step, 4),
                         "step_HBM_GBs_algorithmic": round(ach_step, 1),
                         "step_achieved_TFLOPs": round(fl / (ms_per_step * 1e-3) / 1e12, 1)})
    if by_kernel and not bf16:
        dom_name, dom = max(((k, v) for k, v in by_kernel.items() if v["conv"]), key=lambda kv: kv[1]["ms"])
        achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12 if dom["ms"] > 0 else 0.0
        roofline = {"bound": "mfma", "kernel": dom_name, "achieved": round(achieved, 3), "peak": PEAK_FP32_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_TFLOPS, 4),
                    "traffic": (round(pmc[dom_name]["hbm_bytes_per_launch"]) if dom_name in pmc else None),
                    # PMC counters cannot be read from inside this process: the figure is the committed result of the
                    # separate rocprofv3 --pmc passes of this same command (tools/pmc_traffic.sh), not of this run
                    "traffic_source": ("profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                       "this command, committed; not measured by this run)" if dom_name in pmc else None),
                    "algorithmic_bytes_per_launch": round(dom["bytes"] / dom["launches"]),
                    "launches": dom["launches"], "avg_launch_ms": round(dom["ms"] / dom["launches"], 4),
                    "kernel_share_of_step": round(dom["ms"] / max(all_ms, 1e-9), 3),
                    # whole-step view: sum over the 58 layers of max(F/P, B/BW) vs measured
                    "step_T_roof_ms": round(t_roof_ms / prof_steps, 3),
                    "step_frac_of_fp32_roofline": round((t_roof_ms / prof_steps) / ms_per_step, 4),
                    "step_algorithmic_TFLOPs": round(conv_flops / prof_steps / 1e12, 3),
                    "step_HBM_GBs_algorithmic": round(conv_bytes / prof_steps / 1e9 / (ms_per_step * 1e-3), 1),
                    "kernel_timing": "HIP events per launch, %d steps with the weight-gradient stream serialised "
                                     "(%.1f ms/step serial vs %.1f overlapped)" % (prof_steps, serial_ms_per_step, ms_per_step)}
    if by_kernel and not bf16 and roofline is not None:
        fam = group_families(by_kernel)
        low = [(k, v) for k, v in fam.items() if v["conv"] and v["ms"] >= 0.02 * all_ms]
        if low:
            ln, lv = min(low, key=lambda kv: kv[1]["roof_ms"] / kv[1]["ms"])
            roofline["furthest_below_roofline"] = {
                "kernel": ln + "*", "ms_per_step": round(lv["ms"] / prof_steps, 3), "launches_per_step": lv["launches"] // prof_steps,
                "frac_of_own_roofline": round(lv["roof_ms"] / lv["ms"], 4),
                "achieved_TFLOPs": round(lv["flops"] / (lv["ms"] * 1e-3) / 1e12, 1)}
        roofline["families_ms_per_step"] = {k: round(v["ms"] / prof_steps, 3) for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])}
    if rank == 0 and args.breakdown:
        for k, v in sorted(by_kernel.items(), key=lambda kv: -kv[1]["ms"]):
            tf = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 and v["flops"] else 0.0
            gb = v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] > 0 else 0.0
            sys.stderr.write("%-24s %8.2f ms/step %6d launches/step %8.2f TFLOP/s %9.1f GB/s(alg)\n" % (
                k, v["ms"] / prof_steps, v["launches"] // prof_steps, tf, gb))
        sys.stderr.write("sum of timed launches %.2f ms/step (serialised pass %.2f ms/step), wall %.2f ms/step\n" % (
            all_ms / prof_steps, serial_ms_per_step, ms_per_step))

    if rank == 0 and args.layers:
        pn = ["fwd", "dgrad", "wgrad", "bn_stats", "bn_act", "bn_bwd", "head"]
        # last column: the launch's own roofline time max(F / P, B / BW) over its measured time
        for (lname, ps, k), (ms, fl, by) in sorted(by_layer.items(), key=lambda kv: -kv[1][0])[:400]:
            ms /= prof_steps
            t_roof = max(fl / ((PEAK_BF16_TFLOPS if bf16 else PEAK_FP32_TFLOPS) * 1e12), by / (PEAK_HBM_GBS * 1e9)) * 1e3
            sys.stderr.write("%-52s %-8s %-20s %8.3f ms %8.2f TFLOP/s %8.1f GB/s %5.0f%%\n" % (
                lname, pn[ps], k, ms, fl / (ms * 1e-3) / 1e12 if ms > 0 else 0,
                by / (ms * 1e-3) / 1e9 if ms > 0 else 0, 100.0 * t_roof / ms if ms > 0 else 0))
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and len(dims) == 4:
        cpu = run_cpu_baseline(int(dims[0]))

    secondary = None
    sec_wl = args.secondary_workload or ("cfg5_3d256_f8_b4_bf16" if args.workload == "cfg3_3d192_f8_b4" else "")
    want_secondary = bool(sec_wl) and not args.no_secondary
    if want_secondary:
        # every rank releases its workspace; with several ranks the process group ends here, so that the child job below finds
        # N idle GPUs and no live communicator of this job
        del net
        torch.cuda.empty_cache()
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
    if rank == 0 and want_secondary:
        # BASELINE.json configs[4] (3-D 256^3 bf16, "8 x MI355X") timed by the same invocation at the same N: a CHILD job started
        # after this one has released its workspace (never an exec from a process that holds the GPU).  With N > 1 the child is
        # `bench.py --gpus N`, which launches its own N ranks (torch.distributed.run) before any of them touches a GPU; the
        # launcher variables of THIS rank must not leak into it.  The headline above is untouched.
        try:
            cmd = [sys.executable, os.path.abspath(__file__), "--workload", sec_wl, "--steps", str(args.steps),
                   "--warmup", str(args.warmup), "--no-cpu-baseline", "--gpus", str(world), "--backend", args.backend]
            cenv = {k: v for k, v in os.environ.items()
                    if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "ROLE_WORLD_SIZE",
                                 "MASTER_PORT", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS",
                                 "GROUP_WORLD_SIZE", "ROLE_NAME", "TORCHELASTIC_USE_AGENT_STORE", "TORCH_NCCL_ASYNC_ERROR_HANDLING",
                                 "TORCHELASTIC_ERROR_FILE")}
            cp = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, timeout=900, env=cenv)
            line = [x for x in cp.stdout.strip().split("\n") if x.startswith("{")]
            if cp.returncode == 0 and line:
                c5 = json.loads(line[-1])
                secondary = {"workload": c5["config"]["workload"], "dtype": c5["dtype"], "metric": c5["metric"], "value": c5["value"],
                             "unit": c5["unit"], "n_gpus": c5["n_gpus"], "steps": c5["steps"], "warmup": c5["warmup"],
                             "ms_per_step": c5["ms_per_step"], "config": c5["config"], "last_metrics": c5["last_metrics"],
                             "ranks_seen": c5.get("ranks_seen"), "per_rank": c5.get("per_rank"), "step_parts": c5.get("step_parts"),
                             "roofline": c5["roofline"]}
            else:
                secondary = {"workload": sec_wl, "error": "child exited with %d" % cp.returncode}
        except Exception as e:   # the headline must survive a failing secondary leg
            secondary = {"workload": sec_wl, "error": repr(e)}

    if rank == 0:
        out = {
            "metric": "fwd+bwd images/sec on 3D 192^3x1 U-ResNet" if args.workload.startswith("cfg3")
                      else "fwd+bwd images/sec (%s)" % args.workload,
            "value": round(value, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if bf16 else "f32", "data": "synthetic",
            "config": {"workload": args.workload, "dims": list(dims), "base_filters": base, "num_class": ncls,
                       "batch_per_gpu": batch, "global_batch": batch * world,
                       "step": "zero_gradients+accum_gradients(fwd+loss+bwd)+allreduce+adam",
                       "parallelism": "dp%d" % world, "backend": (args.backend if world > 1 else None)},
            "ranks_seen": ranks_seen, "per_rank": per_rank,
            "last_metrics": {"loss": metrics[0], "acc_all": metrics[1], "acc_nonzero": metrics[2]},
            "step_parts": parts, "host_feed": host_feed,
            "roofline": roofline, "cpu_baseline": cpu, "secondary": secondary,
        }
        print(json.dumps(out))
    if world > 1 and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
