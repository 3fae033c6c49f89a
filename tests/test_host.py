"""CPU tests of the host-side mirror of the reference interface (no GPU, no compute calls)."""
import ctypes
import io
import os
import re
import sys
from contextlib import redirect_stdout

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

import uresnet_amd  # noqa: E402,F401
from uresnet_amd import _lib, ssnet_base, ssnet_config, uresnet  # noqa: E402
from uresnet_amd import synthetic_io as sio  # noqa: E402


# ---- config (lib/config.py) --------------------------------------------------------------------------------
def test_config_defaults_match_reference():
    c = ssnet_config()
    assert (c.NUM_CLASS, c.BASE_NUM_FILTERS, c.LEARNING_RATE, c.MINIBATCH_SIZE, c.NUM_MINIBATCHES) == (3, 16, -1, 10, 5)
    assert (c.TF_RANDOM_SEED, c.TRAIN, c.USE_WEIGHTS, c.REPORT_STEPS, c.CHECKPOINT_NHOUR) == (1234, True, True, 200, 0.4)
    assert c.KEYWORD_DATA == 'data' and c.AVOID_LOAD_PARAMS == []


def test_config_override_and_errors(tmp_path):
    p = tmp_path / "a.cfg"
    p.write_text("NUM_CLASS 5  # comment\n  BASE_NUM_FILTERS\t8\nLEARNING_RATE 0.0001\nSAVE_FILE 'x/y'\n"
                 "AVOID_LOAD_PARAMS ['a', 'b']\nDUMP_IMAGE False\nthis line has three words\n\nTRAIN False\n")
    c = ssnet_config()
    with redirect_stdout(io.StringIO()) as out:
        c.override(str(p))
    assert (c.NUM_CLASS, c.BASE_NUM_FILTERS, c.LEARNING_RATE, c.SAVE_FILE, c.TRAIN) == (5, 8, 0.0001, 'x/y', False)
    assert c.AVOID_LOAD_PARAMS == ['a', 'b']
    assert 'Ignoring a parameter in file: DUMP_IMAGE' in out.getvalue()   # reference crashes here (lib/config.py:57-61)
    assert ssnet_config().NUM_CLASS == 3                                    # instance override, class defaults intact
    bad = tmp_path / "b.cfg"
    bad.write_text("NUM_CLASS 'three'\n")
    with redirect_stdout(io.StringIO()), pytest.raises(TypeError):
        ssnet_config().override(str(bad))
    with redirect_stdout(io.StringIO()), pytest.raises(IOError):
        ssnet_config().override(str(tmp_path / "missing.cfg"))


def test_shipped_configs_parse():
    for name in ["train3d.cfg", "train2d.cfg", "ana3d.cfg"]:
        c = ssnet_config()
        with redirect_stdout(io.StringIO()):
            c.override(os.path.join(ROOT, "config", name))
            c.dump()
    assert c.TRAIN is False and c.BASE_NUM_FILTERS == 8


# ---- topology / debug prints (lib/uresnet.py:22-123) ------------------------------------------------------------
def test_debug_print_sequence_3d():
    net = uresnet(dims=[128, 128, 128, 1], num_class=3, debug=True)   # the reference's own smoke shape (:128-130)
    with redirect_stdout(io.StringIO()) as out:
        net.construct(trainable=True, use_weight=True, allocate=False)
    lines = out.getvalue().strip().split("\n")
    tags = [l.split(") ", 1)[1] for l in lines]
    want = ['input shape', 'after conv0'] + ['after resnet_module%d' % i for i in range(5)]
    for i in range(5):
        want += ['after deconv%d' % i, 'after concat%d' % i, 'after resnet_module%d' % (i + 5)]
    want += ['after conv1', 'after conv2']
    assert tags == want
    assert lines[1].startswith("(-1, 128, 128, 128, 16)") and lines[6].startswith("(-1, 4, 4, 4, 512)")
    assert lines[-1].startswith("(-1, 128, 128, 128, 3)")


def test_parameter_layout_and_counts():
    for dims, F, ncls, want in [([192, 192, 192, 1], 8, 3, 12468083), ([256, 256, 1], 16, 3, 16858979),
                                ([512, 512, 1], 16, 5, 16859269)]:
        net = uresnet(dims=dims, num_class=ncls, base_num_outputs=F)
        net.construct(allocate=False)
        assert net._n_params == want and len(net._specs) == 116
    names = net.variable_names()
    assert names[:2] == ['UResNet/conv0/weights', 'UResNet/conv0/BatchNorm/beta']
    assert names[-2:] == ['UResNet/conv2/weights', 'UResNet/conv2/BatchNorm/beta']
    deconv = [s for s in net._specs if s[0] == 'UResNet/deconv0/weights'][0]
    assert deconv[1] == (3, 3, 256, 512)          # [k,k,Cout,Cin]


def test_reference_error_behaviour():
    with redirect_stdout(io.StringIO()), pytest.raises(NotImplementedError):
        ssnet_base(dims=[4, 4], num_class=3)                       # lib/ssnet.py:12-14
    with pytest.raises(NotImplementedError):
        ssnet_base(dims=[32, 32, 1], num_class=3).construct(allocate=False)   # abstract _build, lib/ssnet.py:17-18

    class other(uresnet):
        def _build(self, input_tensor):
            from uresnet_amd.resnet_module import conv
            return conv(input_tensor, self._num_class, 3, 1, 'only')
    with pytest.raises(NotImplementedError):
        other(dims=[32, 32, 1], num_class=3).construct(allocate=False)
    net = uresnet(dims=[32, 32, 1], num_class=3)
    net.construct(trainable=True, use_weight=True, allocate=False)
    net._params = object()
    with pytest.raises(TypeError):                                  # lib/ssnet.py:143-145
        net.feed_dict(np.zeros((1, 1024), np.float32), np.zeros((1, 1024), np.float32), None)
    assert net._opt._lr == 0.001                                     # AdamOptimizer() default when lr <= 0
    net2 = uresnet(dims=[32, 32, 1], num_class=3)
    net2.construct(trainable=True, use_weight=False, learning_rate=1e-4, allocate=False)
    assert net2._opt._lr == 1e-4


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    net = uresnet(dims=[32, 32, 1], num_class=3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net.construct(trainable=True, use_weight=False)


# ---- C-ABI ---------------------------------------------------------------------------------------------------
def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "uresnet_hip.h")).read()
    declared = set(re.findall(r"\b(ursn_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), name
    assert set(_lib.EXPORTS) <= declared
    assert lib.ursn_abi_version() == _lib.ABI_VERSION


def test_query_sizes_and_errors():
    lib = _lib.load()
    net = uresnet(dims=[192, 192, 192, 1], num_class=3, base_num_outputs=8)
    net.construct(allocate=False)
    cfg = net._native_config(4)
    s = _lib.ursn_sizes()
    assert lib.ursn_query(ctypes.byref(cfg), ctypes.byref(s)) == 0
    assert (s.n_params, s.n_layers, s.n_tensors) == (12468083, 58, 116)
    assert 20e9 < s.workspace_bytes < 80e9
    cfg.spatial[0] = 100                                # not divisible by 32: deconv/skip shapes would differ
    assert lib.ursn_query(ctypes.byref(cfg), ctypes.byref(s)) != 0
    assert b"divisible" in lib.ursn_last_error()
    assert lib.ursn_conv_forward(None, None, None, None, None) != 0 and b"null" in lib.ursn_last_error()
    d = _lib.ursn_conv_desc()
    d.ndim, d.n, d.cin, d.cout, d.k, d.stride = 3, 4, 16, 16, 3, 1
    d.in_sp[0] = d.in_sp[1] = d.in_sp[2] = 96
    assert lib.ursn_conv_wgrad_scratch_bytes(ctypes.byref(d)) > 0


def test_buffer_path_kernels_refuse_planes_beyond_their_offset_range():
    """The z-marching tiled kernels stage through buffer instructions with 32-bit byte offsets inside one z plane and an
    out-of-range marker at 2 GB (csrc/buffer_stage.h): the host plan must hand a layer whose plane reaches the marker to the
    generic kernels instead (plan query only, nothing runs)."""
    lib = _lib.load()
    d = _lib.ursn_conv_desc()
    d.ndim, d.n, d.cin, d.cout, d.k, d.stride = 3, 1, 8, 8, 3, 1
    d.in_sp[0] = 8
    d.in_sp[1] = d.in_sp[2] = 4096          # 4096^2 x 8 channels x 4 B = 512 MB per plane
    assert lib.ursn_conv_bs_blocks(ctypes.byref(d)) > 0
    d.in_sp[1] = d.in_sp[2] = 8192          # 2 GB per plane: at the marker
    assert lib.ursn_conv_bs_blocks(ctypes.byref(d)) == 0


def test_bf16_dispatch_plan_of_the_round4_kernels():
    """Which bf16 kernel family a layer pass is handed to (ursn_conv_plan: plan query only, nothing runs): the single-launch
    stride-2 scatter kernel for the transposed convs / stride-2 data gradients of levels 1-5 (even extents, >= 32 contraction
    channels), the conv0 kernels for one input channel, the deep-level weight gradient for >= 64 produced channels at <= 2^18
    voxels -- and the generic kernels where a guard fails (odd extents, 32 produced channels, a million voxels)."""
    lib = _lib.load()
    buf = ctypes.create_string_buffer(32)

    def plan(ndim, n, sp, ci, co, k, stride, tr, ps):
        d = _lib.ursn_conv_desc()
        d.ndim, d.n, d.cin, d.cout, d.k, d.stride, d.transposed, d.dtype = ndim, n, ci, co, k, stride, tr, 1
        for i, v in enumerate(sp):
            d.in_sp[i] = v
        _lib.check(lib.ursn_conv_plan(ctypes.byref(d), ps, buf, 32))
        return buf.value.decode()

    assert plan(3, 4, (8, 8, 8), 256, 128, 3, 2, 1, 0) == "bsconv"        # deconv0 forward
    assert plan(3, 4, (64, 64, 64), 32, 16, 3, 2, 1, 0) == "bsconv"       # deconv3 forward
    assert plan(3, 4, (128, 128, 128), 16, 8, 3, 2, 1, 0) == "bdeconv"    # deconv4: the 16 -> 8 kernel of round 3
    assert plan(3, 4, (128, 128, 128), 16, 32, 3, 2, 0, 1) == "bsconv"    # stride-2 conv 16 -> 32: its data gradient
    assert plan(3, 1, (7, 9, 21), 16, 32, 3, 2, 0, 1) == "bconv"          # odd extents: by parity class on the box kernel
    assert plan(3, 4, (256, 256, 256), 8, 16, 3, 2, 0, 0) == "bs2k8"     # first stride-2 conv: z-marching gather 8 -> 16
    assert plan(3, 4, (128, 128, 128), 16, 8, 3, 2, 1, 1) == "bs2k8"     # last transposed conv: its data gradient is the same gather
    assert plan(3, 4, (256, 256, 256), 8, 16, 3, 2, 0, 2) == "bs2k8w"    # ... and its weight gradient
    assert plan(3, 4, (256, 256, 256), 1, 8, 3, 1, 0, 0) == "b0conv"
    assert plan(3, 4, (256, 256, 256), 1, 8, 3, 1, 0, 2) == "b0wgrad"
    assert plan(3, 4, (32, 32, 32), 64, 64, 3, 1, 0, 2) == "bdwgrad"
    assert plan(3, 4, (8, 8, 8), 256, 256, 3, 1, 0, 2) == "bdwgrad"
    assert plan(3, 4, (32, 32, 32), 128, 64, 3, 1, 0, 2) == "bdwgrad"     # decoder conv1: 2C -> C
    assert plan(3, 4, (64, 64, 64), 32, 32, 3, 1, 0, 2) == "bwgrad"       # 32 produced channels / a million voxels: generic
    assert plan(3, 4, (32, 32, 32), 64, 64, 3, 1, 0, 0) == "bdconv"
    assert plan(3, 4, (256, 256, 256), 8, 8, 3, 1, 0, 0) == "b3conv"


# ---- synthetic IO (larcv_threadio protocol) ------------------------------------------------------------------------
def test_synthetic_threadio_protocol():
    io_ = sio.synthetic_threadio()
    io_.configure({'filler_name': 'MainIO', 'verbosity': 0,
                   'filler_cfg': {'Dims': [32, 32, 32, 1], 'NumClass': 3, 'Generator': 'lartpc_sparse'}})
    io_.start_manager(2)
    io_.next(store_entries=True, store_event_ids=True)
    d = io_.fetch_data('data')
    assert d.dim() == [2, 32, 32, 32, 1] and d.data().shape == (2, 32768) and d.data().dtype == np.float32
    assert io_.fetch_data('label').dim() == [2, 32, 32, 32]
    lab, w = io_.fetch_data('label').data(), io_.fetch_data('weight').data()
    assert set(np.unique(lab)) <= {0.0, 1.0, 2.0} and (lab > 0).any()
    assert np.all((d.data() > 0) == (lab > 0)) and d.data().max() <= 500 and d.data()[d.data() > 0].min() >= 1
    assert io_.fetch_entries() == [0, 1]
    io_.next()
    assert io_.fetch_entries() == [2, 3]
    again = sio.lartpc_sparse([32, 32, 32, 1], 3, 2)[0]
    assert np.array_equal(again, io_.fetch_data('data').data()[0])       # entry k is reproducible anywhere
    with pytest.raises(KeyError):
        io_.fetch_data('nope')
    io_.reset()


def test_rank_sharding_of_entries():
    seen = []
    for rank in range(2):
        io_ = sio.synthetic_threadio()
        io_.configure({'filler_cfg': {'Dims': [16, 16, 1], 'NumClass': 3, 'Generator': 'dense_uniform',
                                      'FirstEntry': rank, 'EntryStride': 2}})
        io_.start_manager(3)
        io_.next()
        seen.append(io_.fetch_entries())
        io_.reset()
    assert seen == [[0, 2, 4], [1, 3, 5]]


def test_num_strides_above_five_rejected():
    """lib/uresnet.py:100 names the decoder units 'resnet_module%d' % (step + 5): a sixth stride would collide with
    the encoder's variable scopes (TensorFlow raises there too), and 10 tensors would silently share checkpoint keys."""
    with pytest.raises(ValueError, match="num_strides"):
        uresnet(dims=[64, 64, 1], num_class=3, num_strides=6).construct(allocate=False)
    lib = _lib.load()
    net = uresnet(dims=[64, 64, 1], num_class=3, num_strides=5)
    net.construct(allocate=False)
    assert len(set(net.variable_names())) == len(net.variable_names()) == 116
    cfg = net._native_config(1)
    cfg.num_strides = 6
    s = _lib.ursn_sizes()
    assert lib.ursn_query(ctypes.byref(cfg), ctypes.byref(s)) != 0 and b"num_strides" in lib.ursn_last_error()


def test_plan_check_reads_the_native_tables():
    """construct() compares what _build recorded with the layer / concat tables of the compiled plan (ursn_query_layer,
    ursn_query_concat): a subclass that changes an activation or swaps the tf.concat operands (lib/uresnet.py:81) is
    refused instead of silently running the fixed native topology."""
    U = sys.modules["uresnet_amd.uresnet"]
    lib = _lib.load()
    net = uresnet(dims=[64, 64, 64, 1], num_class=3, base_num_outputs=8)
    net.construct(allocate=False)
    info = _lib.ursn_layer_info()
    relu_layers = []
    for i in range(58):
        assert lib.ursn_query_layer(ctypes.byref(net._cfg), i, ctypes.byref(info)) == 0
        if info.relu:
            relu_layers.append(info.name.decode())
    assert relu_layers == ['UResNet/conv0'] + ['UResNet/deconv%d' % i for i in range(5)] + ['UResNet/conv1']
    assert lib.ursn_query_layer(ctypes.byref(net._cfg), 58, ctypes.byref(info)) != 0
    a, b = ctypes.create_string_buffer(128), ctypes.create_string_buffer(128)
    cats = []
    for i in range(5):
        assert lib.ursn_query_concat(ctypes.byref(net._cfg), i, a, b, 128) == 0
        cats.append((a.value.decode(), b.value.decode()))
    assert cats[0] == ('UResNet/deconv0', 'UResNet/resnet_module3/module2')     # concat0 <- module3
    assert cats[4] == ('UResNet/deconv4', 'UResNet/conv0')
    assert [(x, y) for _, x, y in net._graph.concats] == cats

    class swapped(uresnet):
        def _build(self, input_tensor):
            orig = U.concat
            U.concat = lambda ts, name: orig(ts[::-1], name)
            try:
                return uresnet._build(self, input_tensor)
            finally:
                U.concat = orig
    with pytest.raises(NotImplementedError, match="concat"):
        swapped(dims=[64, 64, 1], num_class=3).construct(allocate=False)

    class relu_logits(uresnet):
        def _build(self, input_tensor):
            orig = U.conv

            def conv(inputs, n, k, s, scope, activation_fn=None):
                return orig(inputs, n, k, s, scope, activation_fn='relu' if scope == 'conv2' else activation_fn)
            U.conv = conv
            try:
                return uresnet._build(self, input_tensor)
            finally:
                U.conv = orig
    with pytest.raises(NotImplementedError, match="conv2"):
        relu_logits(dims=[64, 64, 1], num_class=3).construct(allocate=False)


def test_native_plan_against_reference_graph():
    """The layer / concat tables compiled into liburesnet_hip.so (ursn_query_layer / ursn_query_concat) against the
    reference's saved GraphDef (tests/golden/ref_graph.json): scope order, conv vs transposed conv, strides, channel
    counts, [deconv_i, skip] operand order.  (conv0 / conv1 were 7x7 in that older revision: kernel size excluded.)"""
    import json
    with open(os.path.join(ROOT, "tests", "golden", "ref_graph.json")) as f:
        G = json.load(f)
    lib = _lib.load()
    net = uresnet(dims=[512, 512, 1], num_class=3, base_num_outputs=16)
    net.construct(allocate=False)
    info = _lib.ursn_layer_info()
    for i, c in enumerate(G["forward_convs"]):
        assert lib.ursn_query_layer(ctypes.byref(net._cfg), i, ctypes.byref(info)) == 0
        assert info.name.decode() == c["scope"]
        assert bool(info.transposed) == (c["op"] == "Conv2DBackpropInput")
        assert c["strides"] == [1, info.stride, info.stride, 1]
        fs = c["filter_shape"]
        assert (fs[2], fs[3]) == ((info.cout, info.cin) if info.transposed else (info.cin, info.cout))
        if c["scope"] not in ("UResNet/conv0", "UResNet/conv1"):
            assert fs[0] == fs[1] == info.k
    a, b = ctypes.create_string_buffer(128), ctypes.create_string_buffer(128)
    for i, c in enumerate(G["concats"]):
        assert lib.ursn_query_concat(ctypes.byref(net._cfg), i, a, b, 128) == 0
        assert [a.value.decode(), b.value.decode()] == c["inputs"]
    # initialize_variables draws inside the graph's Xavier bounds
    for name, shape, off, n in net._specs:
        if name.endswith("/weights") and name.rsplit("/", 1)[0] not in ("UResNet/conv0", "UResNet/conv1"):
            lim = G["xavier_limits"][name.rsplit("/", 1)[0]]["limit"]
            fan = int(np.prod(shape[:2]))
            assert abs(np.sqrt(6.0 / (fan * (shape[-1] + shape[-2]))) - lim) < 1e-8


def test_library_is_built_from_the_sources_as_they_are():
    """The in-tree .so is reused only while the digest written beside it at build time matches the sources (a stale library
    with fresh timestamps must rebuild, not pass an mtime comparison)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("_ursn_build_t", os.path.join(root, "u-resnet_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    b.build(force=False, verbose=False)
    assert os.path.isfile(b.LIB) and not b.needs_build()
    with open(b.STAMP) as f:
        assert f.read().strip() == b.sources_digest()
    real = b.sources_digest
    try:
        b.sources_digest = lambda: "0" * 64     # any source change
        assert b.needs_build()
    finally:
        b.sources_digest = real
