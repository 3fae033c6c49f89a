"""Python 3 mirror of the reference plugin surface lib/ssnet.py (class ssnet_base).

Same constructor, same abstract ``_build`` hook, same ``construct`` signature and the same six
run methods with the same return structures (lib/ssnet.py:91-153).  Where the reference builds a
TensorFlow graph and each method is one ``sess.run`` fetch-set, here ``construct`` records the
topology symbolically, checks it against the launch plan compiled into liburesnet_hip.so and
each method is one call through the C-ABI (include/uresnet_hip.h).  PyTorch-ROCm only provides
device memory, streams and the RCCL process group.  There is no CPU fallback.
"""
from __future__ import print_function

import ctypes
import math
import sys

import numpy as np

from . import _lib
from .resnet_module import Graph, SymTensor


class HipSession(object):
    """Stands in for ``tf.Session`` (lib/ssnet_trainval.py:112): names the device and stream the
    fetch-sets run on.  Every run method also accepts ``sess=None`` (current device/stream)."""

    def __init__(self, device=None):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError('HipSession: no HIP device visible; the U-ResNet path has no CPU fallback')
        self.device = torch.device('cuda', torch.cuda.current_device() if device is None else int(device))

    def stream_ptr(self):
        import torch
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)


class _Adam(object):
    """Carries ``_lr`` like tf.train.AdamOptimizer (read at lib/ssnet_trainval.py:209)."""

    def __init__(self, lr):
        self._lr = lr
        self._beta1, self._beta2, self._epsilon = 0.9, 0.999, 1e-8


class ssnet_base(object):

    def __init__(self, dims, num_class):
        self._dims = np.array(dims, np.int32)
        if not len(self._dims) in [3, 4]:
            print('Error: len(dims) =', len(self._dims), 'but only 3 (H,W,C) or 4 (H,W,D,C) supported!')
            raise NotImplementedError
        self._num_class = int(num_class)
        self._handle = None
        self._max_batch = 0
        self._params = None

    def _build(self, input_tensor):
        raise NotImplementedError

    # ------------------------------------------------------------------------------------------
    # construct (lib/ssnet.py:20-89)
    # ------------------------------------------------------------------------------------------
    def construct(self, trainable=True, use_weight=True, learning_rate=None, allocate=True, device=None,
                  seed=1234, bn_eps=1e-3, precision='fp32'):
        """``precision='bf16'`` (not in the reference, which is fp32 TensorFlow): mixed precision -- activations and
        gradient tensors bf16 in HBM, bf16 MFMA convolutions with fp32 accumulation; parameters, BatchNorm statistics,
        accumulated gradients and Adam stay fp32 (BASELINE.json configs[4])."""
        if precision not in ('fp32', 'bf16'):
            raise ValueError("precision must be 'fp32' or 'bf16'")
        self._precision = precision
        self._trainable = bool(trainable)
        self._use_weight = bool(use_weight)
        self._learning_rate = learning_rate
        self._bn_eps = float(bn_eps)

        self._data_size = int(np.prod(self._dims))
        self._label_size = int(np.prod(self._dims[:-1]))

        graph = Graph()
        shape_dim = tuple(int(d) for d in np.insert(self._dims, 0, -1))
        net = self._build(input_tensor=SymTensor(shape_dim, graph, 'input_prep/data_reshape'))
        if tuple(net.shape[1:]) != tuple(shape_dim[1:-1]) + (self._num_class,):
            raise ValueError('_build returned logits of shape %s' % (net.shape,))
        self._graph = graph

        self._softmax = None
        self._loss = None
        self._accuracy_allpix = None
        self._accuracy_nonzero = None
        self._merged_summary = None
        self._opt = None
        if self._trainable:
            # lib/ssnet.py:72-75: default AdamOptimizer() when learning_rate <= 0
            if self._learning_rate is None or self._learning_rate <= 0:
                self._opt = _Adam(0.001)
            else:
                self._opt = _Adam(self._learning_rate)

        self._cfg = self._native_config(max_batch=1)
        self._check_plan()
        if allocate:
            self._allocate(device, seed)

    def _native_config(self, max_batch):
        cfg = _lib.ursn_config()
        cfg.ndim = len(self._dims) - 1
        for i in range(3):
            cfg.spatial[i] = int(self._dims[i]) if i < cfg.ndim else 1
        cfg.cin = int(self._dims[-1])
        cfg.base_filters = int(getattr(self, '_base_num_outputs', 16))
        cfg.num_class = self._num_class
        cfg.num_strides = int(getattr(self, '_num_strides', 5))
        cfg.max_batch = int(max_batch)
        cfg.trainable = int(self._trainable)
        cfg.use_weight = int(self._use_weight)
        cfg.bn_eps = self._bn_eps
        cfg.act_dtype = 1 if getattr(self, '_precision', 'fp32') == 'bf16' else 0
        return cfg

    def _check_plan(self):
        """The native library implements the U-ResNet topology of lib/uresnet.py; a subclass whose
        ``_build`` records anything else cannot be executed (there is no generic graph executor).
        What ``_build`` recorded -- every conv-like layer's scope, kind, kernel, stride, channels and
        activation, and the operand order of every tf.concat -- is compared with the layer / concat
        tables of the plan the library compiles for this configuration (ursn_query_layer /
        ursn_query_concat), not with a second Python copy of the formula."""
        lib = _lib.load()
        if int(getattr(self, '_num_strides', 5)) > 5:
            # lib/uresnet.py:100 names the decoder units 'resnet_module%d' % (step + 5): with more than 5 strides they
            # collide with the encoder's scopes (TensorFlow raises on the duplicate variable scope as well)
            raise ValueError('num_strides > 5: decoder scope names collide with the encoder (lib/uresnet.py:100)')
        sizes = _lib.ursn_sizes()
        _lib.check(lib.ursn_query(ctypes.byref(self._cfg), ctypes.byref(sizes)))
        self._n_params = int(sizes.n_params)
        native = []
        info = _lib.ursn_layer_info()
        for i in range(int(sizes.n_layers)):
            _lib.check(lib.ursn_query_layer(ctypes.byref(self._cfg), i, ctypes.byref(info)))
            native.append(dict(name=info.name.decode(), kind='deconv' if info.transposed else 'conv', k=int(info.k),
                               stride=int(info.stride), cin=int(info.cin), cout=int(info.cout), relu=bool(info.relu),
                               w_offset=int(info.w_offset), beta_offset=int(info.beta_offset)))
        keys = ('name', 'kind', 'k', 'stride', 'cin', 'cout', 'relu')
        got = [tuple(l[k] for k in keys) for l in self._graph.layers]
        want = [tuple(l[k] for k in keys) for l in native]
        if got != want:
            diff = [(g, w) for g, w in zip(got, want) if g != w][:1] or [(len(got), len(want))]
            raise NotImplementedError('_build recorded a topology other than lib/uresnet.py:22-123 (first difference, '
                                      'recorded vs native plan: %r); only that hot path is implemented natively' % (diff[0],))
        a, b = ctypes.create_string_buffer(128), ctypes.create_string_buffer(128)
        ns = int(getattr(self, '_num_strides', 5))
        native_cats = []
        for step in range(ns):
            _lib.check(lib.ursn_query_concat(ctypes.byref(self._cfg), step, a, b, 128))
            native_cats.append((a.value.decode(), b.value.decode()))
        if [(x, y) for (_, x, y) in self._graph.concats] != native_cats:
            raise NotImplementedError('_build concatenates %r; the native plan implements tf.concat([deconv_i, skip]) = %r '
                                      '(lib/uresnet.py:81)' % ([(x, y) for (_, x, y) in self._graph.concats], native_cats))
        names = [l['name'] for l in native]
        if len(set(names)) != len(names):
            raise ValueError('duplicate variable scopes in the plan')
        specs = []
        nd = len(self._dims) - 1
        for l in native:
            wshape = (l['k'],) * nd + ((l['cin'], l['cout']) if l['kind'] == 'conv' else (l['cout'], l['cin']))
            specs.append((l['name'] + '/weights', wshape, l['w_offset'], int(np.prod(wshape))))
            specs.append((l['name'] + '/BatchNorm/beta', (l['cout'],), l['beta_offset'], l['cout']))
        end = max(off + n for _, _, off, n in specs)
        if end != self._n_params or sum(n for _, _, _, n in specs) != self._n_params:
            raise RuntimeError('parameter table does not tile the flat buffer: %d vs native %d' % (end, self._n_params))
        self._specs = specs

    # ------------------------------------------------------------------------------------------
    # device state
    # ------------------------------------------------------------------------------------------
    def _allocate(self, device, seed):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError('ssnet_base.construct: no HIP device visible; the U-ResNet path has no CPU fallback '
                               '(pass allocate=False to only record/check the topology)')
        self._device = torch.device('cuda', torch.cuda.current_device() if device is None else int(device))
        n = self._n_params
        self._params = torch.empty(n, dtype=torch.float32, device=self._device)
        self._grads = self._adam_m = self._adam_v = None
        if self._trainable:
            self._grads = torch.zeros(n, dtype=torch.float32, device=self._device)
            self._adam_m = torch.zeros(n, dtype=torch.float32, device=self._device)
            self._adam_v = torch.zeros(n, dtype=torch.float32, device=self._device)
        self.initialize_variables(seed)

    def initialize_variables(self, seed=1234):
        """tf.global_variables_initializer (lib/ssnet_trainval.py:113): Xavier-uniform weights, beta = 0
        (SURVEY.md Appendix B-6).  TensorFlow's RNG stream is not reproduced."""
        import torch
        rng = np.random.default_rng(seed)
        host = np.zeros(self._n_params, np.float32)
        nd = len(self._dims) - 1
        for name, shape, off, n in self._specs:
            if name.endswith('/weights'):
                fan = int(np.prod(shape[:nd]))
                lim = math.sqrt(6.0 / (fan * (shape[-1] + shape[-2])))
                host[off:off + n] = rng.uniform(-lim, lim, size=n).astype(np.float32)
        self._params.copy_(torch.from_numpy(host))
        if self._trainable:
            self._adam_m.zero_(); self._adam_v.zero_(); self._grads.zero_()
        if self._handle is not None:
            _lib.check(_lib.load().ursn_set_adam_step(self._handle, 0))

    def _ensure_handle(self, batch):
        import torch
        if self._params is None:
            raise RuntimeError('construct(allocate=True) has not been called')
        if self._handle is not None and batch <= self._max_batch:
            return
        lib = _lib.load()
        step = 0
        if self._handle is not None:
            t = ctypes.c_int64(0)
            _lib.check(lib.ursn_get_adam_step(self._handle, ctypes.byref(t)))
            step = t.value
            self._destroy()
        cfg = self._native_config(max_batch=batch)
        sizes = _lib.ursn_sizes()
        _lib.check(lib.ursn_query(ctypes.byref(cfg), ctypes.byref(sizes)))
        # activations of every layer stay resident for the backward pass (sized for 288 GB of HBM, no recomputation): say
        # so before the allocator raises something less readable
        self._workspace = None
        torch.cuda.empty_cache()
        free_b, total_b = torch.cuda.mem_get_info(self._device)
        if int(sizes.workspace_bytes) + (64 << 20) > free_b:
            raise MemoryError('U-ResNet workspace for batch %d needs %.1f GB but only %.1f of %.1f GB of HBM are free; '
                              'use a smaller MINIBATCH_SIZE with more NUM_MINIBATCHES (same gradient sum, '
                              'lib/ssnet.py:77)' % (batch, sizes.workspace_bytes / 1e9, free_b / 1e9, total_b / 1e9))
        self._workspace = torch.empty(int(sizes.workspace_bytes) + 256, dtype=torch.uint8, device=self._device)
        wptr = (self._workspace.data_ptr() + 255) & ~255
        h = ctypes.c_void_p()
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        _lib.check(lib.ursn_create(ctypes.byref(cfg), p(self._params), p(self._grads), p(self._adam_m),
                                   p(self._adam_v), ctypes.c_void_p(wptr), int(sizes.workspace_bytes),
                                   ctypes.byref(h)))
        self._handle, self._max_batch, self._cfg = h, batch, cfg
        _lib.check(lib.ursn_set_adam_step(self._handle, step))

    def _destroy(self):
        if getattr(self, '_handle', None) is not None:
            _lib.load().ursn_destroy(self._handle)
            self._handle = None
            self._workspace = None

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    def _stream(self, sess):
        import torch
        return ctypes.c_void_p(torch.cuda.current_stream(self._device).cuda_stream)

    # ------------------------------------------------------------------------------------------
    # host feed (lib/ssnet.py:141-153 feed_dict; lib/ssnet_trainval.py:167-188 hands over IO buffers that are only
    # valid until the next io.next() and are mutated in place at :173)
    # ------------------------------------------------------------------------------------------
    class _FeedSlot(object):
        """One role's (data / label / weight) staging state: two device buffers used alternately, one pinned host
        buffer for callers that hand over pageable memory, and the events that order copy and compute."""
        __slots__ = ('dev', 'pinned', 'turn', 'consumed', 'copied')

        def __init__(self):
            self.dev, self.pinned, self.turn = [None, None], None, 0
            self.consumed = [None, None]   # recorded on the compute stream after the last launch that reads dev[i]
            self.copied = None

    def _copy_stream(self):
        import torch
        if getattr(self, '_h2d_stream', None) is None:
            self._h2d_stream = torch.cuda.Stream(device=self._device)
            self._feed_slots = {}
            self.feed_stats = {'h2d_bytes': 0, 'h2d_calls': 0, 'staged_bytes': 0}
        return self._h2d_stream

    def _feed(self, x, cols, what):
        """numpy / torch input [N, cols] -> contiguous fp32 device tensor.

        Host arrays travel on a dedicated copy stream from page-locked memory: directly from the caller's buffer when
        it already is page-locked (synthetic_threadio produces into pinned buffers), else through a pinned staging
        buffer.  The call returns once the COPY has completed (the caller may then reuse / mutate its buffer, as the
        reference's IO does), but it never waits for compute: with ``accum_gradients(fetch=False)`` the copy of
        minibatch k+1 overlaps the kernels of minibatch k.  Each role alternates between two device buffers; a buffer
        is only overwritten after the launch that last read it has finished (event recorded by ``_mark_consumed``)."""
        import torch
        if isinstance(x, torch.Tensor):
            if x.device == self._device and x.dtype == torch.float32 and x.is_contiguous():
                return x.reshape(-1, cols)
            if x.device.type != 'cpu':
                return x.to(device=self._device, dtype=torch.float32).reshape(-1, cols).contiguous()
            x = x.numpy()
        host = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, cols)
        n = host.shape[0]
        cs = self._copy_stream()
        slot = self._feed_slots.setdefault(what, ssnet_base._FeedSlot())
        i = slot.turn
        slot.turn ^= 1
        if slot.dev[i] is None or slot.dev[i].shape[0] < n:
            slot.dev[i] = torch.empty((n, cols), dtype=torch.float32, device=self._device)
            slot.consumed[i] = None
            # the caching allocator hands out blocks in compute-stream order: a block freed with kernels still pending on
            # the compute stream may come back here, and the copy stream must not write it before they have finished
            cs.wait_stream(torch.cuda.current_stream(self._device))
            slot.dev[i].record_stream(cs)
        dst = slot.dev[i][:n]
        src = torch.from_numpy(host)
        if not src.is_pinned():
            if slot.pinned is None or slot.pinned.shape[0] < n:
                slot.pinned = torch.empty((n, cols), dtype=torch.float32, pin_memory=True)
            if slot.copied is not None:
                slot.copied.synchronize()      # the previous copy out of the staging buffer
            slot.pinned[:n].copy_(src)
            src = slot.pinned[:n]
            self.feed_stats['staged_bytes'] += host.nbytes
        with torch.cuda.stream(cs):
            if slot.consumed[i] is not None:
                cs.wait_event(slot.consumed[i])
            dst.copy_(src, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(cs)
        slot.copied = ev
        torch.cuda.current_stream(self._device).wait_event(ev)
        ev.synchronize()                       # copy only; kernels of earlier minibatches keep running
        self.feed_stats['h2d_bytes'] += host.nbytes
        self.feed_stats['h2d_calls'] += 1
        return dst

    def _mark_consumed(self, fd):
        """After the launches reading the fed tensors are queued: later copies into the same device buffers wait."""
        import torch
        slots = getattr(self, '_feed_slots', None)
        if not slots:
            return
        ev = None
        for t in fd.values():
            for slot in slots.values():
                for i in (0, 1):
                    if slot.dev[i] is not None and t.data_ptr() == slot.dev[i].data_ptr():
                        if ev is None:
                            ev = torch.cuda.Event()
                            ev.record(torch.cuda.current_stream(self._device))
                        slot.consumed[i] = ev

    # ------------------------------------------------------------------------------------------
    # fetch-sets (lib/ssnet.py:91-153)
    # ------------------------------------------------------------------------------------------
    def feed_dict(self, input_data, input_label=None, input_weight=None):
        if input_weight is None and self._use_weight:
            sys.stderr.write('Network configured to use loss pixel-weighting. Cannot run w/ input_weight=None...\n')
            raise TypeError
        fd = {'input_data': self._feed(input_data, self._data_size, 'data')}
        if input_label is not None:
            fd['input_label'] = self._feed(input_label, self._label_size, 'label')
        if input_weight is not None:
            fd['input_weight'] = self._feed(input_weight, self._label_size, 'weight')
        n = fd['input_data'].shape[0]
        for k, v in fd.items():
            if v.shape[0] != n:
                raise ValueError('%s has batch %d, input_data has %d' % (k, v.shape[0], n))
        return fd

    @staticmethod
    def _ptr(t):
        return ctypes.c_void_p(t.data_ptr()) if t is not None else None

    def make_summary(self, sess, input_data, input_label, input_weight=None):
        """The reference returns a serialized TensorBoard summary; here the three scalars it holds."""
        res, _ = self.run_test(sess, input_data, input_label, input_weight)
        return {'loss': res[0], 'accuracy_all': res[1], 'accuracy_nonzero': res[2]}

    def zero_gradients(self, sess=None):
        if not self._trainable:
            raise RuntimeError('zero_gradients: constructed with trainable=False')
        self._ensure_handle(max(self._max_batch, 1))
        _lib.check(_lib.load().ursn_zero_grad(self._handle, self._stream(sess)))
        return [None]

    def accum_gradients(self, sess, input_data, input_label, input_weight=None, fetch=True):
        if not self._trainable:
            raise RuntimeError('accum_gradients: constructed with trainable=False')
        fd = self.feed_dict(input_data=input_data, input_label=input_label, input_weight=input_weight)
        n = int(fd['input_data'].shape[0])
        self._ensure_handle(n)
        out = (ctypes.c_float * 3)()
        w = fd.get('input_weight') if self._use_weight else None
        _lib.check(_lib.load().ursn_accum_step(self._handle, self._ptr(fd['input_data']), self._ptr(fd['input_label']),
                                               self._ptr(w), n, out if fetch else None, self._stream(sess)))
        self._last_feed = fd  # keep device inputs alive until the stream has consumed them
        self._mark_consumed(fd)
        doc = ['', 'loss', 'acc. all', 'acc. nonzero']
        if not fetch:
            return None, doc
        return [None, float(out[0]), float(out[1]), float(out[2])], doc

    def last_feed(self):
        """Device-resident tensors of the most recent accum_gradients / inference_labels call (valid until two more
        batches have been fed): lets a caller re-run them (summary) without another host copy."""
        return dict(self._last_feed)

    def read_metrics(self, sess=None):
        out = (ctypes.c_float * 3)()
        _lib.check(_lib.load().ursn_read_metrics(self._handle, out, self._stream(sess)))
        return [float(out[0]), float(out[1]), float(out[2])]

    def apply_gradients(self, sess=None):
        if not self._trainable:
            raise RuntimeError('apply_gradients: constructed with trainable=False')
        self.allreduce_gradients()
        self._ensure_handle(max(self._max_batch, 1))
        _lib.check(_lib.load().ursn_apply_adam(self._handle, float(self._opt._lr), self._stream(sess)))
        return [None]

    def allreduce_gradients(self):
        """Data parallelism (not in the reference): every rank accumulated its own minibatches, the
        flat gradient buffer is SUMMED over ranks -- the same reduction as assign_add over
        NUM_MINIBATCHES (lib/ssnet.py:77) -- so N ranks == NUM_MINIBATCHES=N on one device."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self._grads, op=dist.ReduceOp.SUM)

    def run_test(self, sess, input_data, input_label, input_weight=None):
        fd = self.feed_dict(input_data=input_data, input_label=input_label, input_weight=input_weight)
        n = int(fd['input_data'].shape[0])
        self._ensure_handle(n)
        out = (ctypes.c_float * 3)()
        w = fd.get('input_weight') if self._use_weight else None
        _lib.check(_lib.load().ursn_eval(self._handle, self._ptr(fd['input_data']), self._ptr(fd['input_label']),
                                         self._ptr(w), n, out, self._stream(sess)))
        return [float(out[0]), float(out[1]), float(out[2])], ['loss', 'acc. all', 'acc. nonzero']

    def inference(self, sess, input_data, input_label=None, as_numpy=True):
        import torch
        fd = {'input_data': self._feed(input_data, self._data_size, 'data')}
        if input_label is not None:
            fd['input_label'] = self._feed(input_label, self._label_size, 'label')
        n = int(fd['input_data'].shape[0])
        self._ensure_handle(n)
        sm = torch.empty((n,) + tuple(int(d) for d in self._dims[:-1]) + (self._num_class,), dtype=torch.float32,
                         device=self._device)
        out = (ctypes.c_float * 2)()
        _lib.check(_lib.load().ursn_infer(self._handle, self._ptr(fd['input_data']), self._ptr(fd.get('input_label')),
                                          n, self._ptr(sm), out, self._stream(sess)))
        res = [sm.cpu().numpy() if as_numpy else sm]
        if input_label is not None:
            res += [float(out[0]), float(out[1])]
        return res

    def inference_labels(self, sess, input_data, input_label=None, as_numpy=True, with_softmax=False):
        """ana_step's shower/track label volume computed on the device (lib/ssnet_trainval.py:285-287): returns
        [labels [N, *spatial] float32 (, acc_all, acc_nonzero if input_label is given)]; the softmax only leaves the
        kernel when ``with_softmax`` asks for it (appended last, same forward pass)."""
        import torch
        fd = {'input_data': self._feed(input_data, self._data_size, 'data')}
        if input_label is not None:
            fd['input_label'] = self._feed(input_label, self._label_size, 'label')
        n = int(fd['input_data'].shape[0])
        self._ensure_handle(n)
        sp = tuple(int(x) for x in self._dims[:-1])
        out = torch.empty((n,) + sp, dtype=torch.float32, device=self._device)
        sm = torch.empty((n,) + sp + (self._num_class,), dtype=torch.float32, device=self._device) if with_softmax else None
        acc = (ctypes.c_float * 2)()
        _lib.check(_lib.load().ursn_infer_labels(self._handle, self._ptr(fd['input_data']), self._ptr(fd.get('input_label')),
                                                 n, self._ptr(out), self._ptr(sm), acc, self._stream(sess)))
        self._last_feed = fd
        self._mark_consumed(fd)
        res = [out.cpu().numpy() if as_numpy else out]
        if input_label is not None:
            res += [float(acc[0]), float(acc[1])]
        if with_softmax:
            res.append(sm.cpu().numpy() if as_numpy else sm)
        return res

    # ------------------------------------------------------------------------------------------
    # variables (checkpoint / weight injection)
    # ------------------------------------------------------------------------------------------
    def variable_names(self):
        return [s[0] for s in self._specs]

    def get_variables(self):
        host = self._params.detach().cpu().numpy()
        return {name: host[off:off + n].reshape(shape).copy() for name, shape, off, n in self._specs}

    def set_variables(self, values, strict=True):
        import torch
        host = self._params.detach().cpu().numpy().copy()
        for name, shape, off, n in self._specs:
            if name not in values:
                if strict:
                    raise KeyError(name)
                continue
            v = np.asarray(values[name], dtype=np.float32)
            if tuple(v.shape) != tuple(shape):
                raise ValueError('%s: shape %s, expected %s' % (name, v.shape, shape))
            host[off:off + n] = v.reshape(-1)
        self._params.copy_(torch.from_numpy(host))

    def get_gradients(self):
        host = self._grads.detach().cpu().numpy()
        return {name: host[off:off + n].reshape(shape).copy() for name, shape, off, n in self._specs}

    def debug_tensor(self, name):
        """Internal tensor of the last forward/backward as numpy [N, spatial..., C] (parity tests)."""
        import torch
        ptr, vox, ch, cs = ctypes.c_void_p(), ctypes.c_int64(), ctypes.c_int32(), ctypes.c_int32()
        _lib.check(_lib.load().ursn_tensor(self._handle, name.encode(), ctypes.byref(ptr), ctypes.byref(vox),
                                           ctypes.byref(ch), ctypes.byref(cs)))
        from . import hiprt
        if vox.value == 0:   # <scope>:mean / :rstd -- a per-channel fp32 vector in both precisions
            torch.cuda.synchronize(self._device)
            buf = (ctypes.c_float * ch.value)()
            hiprt.memcpy_d2h(buf, ptr.value, ch.value * 4)
            return np.frombuffer(buf, dtype=np.float32).copy()
        n = self._last_feed['input_data'].shape[0]
        total = int(n * vox.value * cs.value)
        valid = total - (cs.value - ch.value)  # a channel-slice view ends `ch` floats into its last voxel
        torch.cuda.synchronize(self._device)
        bf16 = getattr(self, '_precision', 'fp32') == 'bf16'
        esz = 2 if bf16 else 4
        # one pinned staging buffer per net, grown on demand: a pageable ctypes array cost a memset, a slow copy and a
        # second pass for the contiguous result (1.5 s per 900 MB tensor; the full-size in-situ tests fetch forty of them)
        stage = getattr(self, '_dbg_stage', None)
        if stage is None or stage.numel() < total * esz:
            stage = torch.empty(total * esz, dtype=torch.uint8, pin_memory=True)
            self._dbg_stage = stage
        hiprt.memcpy_d2h(ctypes.c_void_p(stage.data_ptr()), ptr.value, valid * esz)
        raw = stage.numpy()[:total * esz]
        shape = (n,) + tuple(int(d) for d in self._level_dims(int(vox.value))) + (ch.value,)
        if bf16:   # bf16 bit patterns: widen on the host (one pass, into the result)
            a = raw.view(np.uint16).reshape(n, int(vox.value), cs.value)[:, :, :ch.value]
            out = np.empty(shape, dtype=np.float32)
            np.left_shift(a.reshape(shape), 16, out=out.view(np.uint32), dtype=np.uint32, casting='unsafe')
            return out
        a = raw.view(np.float32).reshape(n, int(vox.value), cs.value)[:, :, :ch.value]
        return a.reshape(shape).copy()

    def _level_dims(self, voxels):
        sp = [int(d) for d in self._dims[:-1]]
        while int(np.prod(sp)) > voxels:
            sp = [d // 2 for d in sp]
        return sp
