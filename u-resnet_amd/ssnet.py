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
                  seed=1234, bn_eps=1e-3):
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
        return cfg

    def _check_plan(self):
        """The native library implements the U-ResNet topology of lib/uresnet.py; a subclass whose
        ``_build`` records anything else cannot be executed (there is no generic graph executor)."""
        lib = _lib.load()
        sizes = _lib.ursn_sizes()
        _lib.check(lib.ursn_query(ctypes.byref(self._cfg), ctypes.byref(sizes)))
        self._n_params = int(sizes.n_params)
        got = [(l['name'], l['kind'], l['k'], l['stride'], l['cin'], l['cout']) for l in self._graph.layers]
        want = self._expected_layers()
        if got != want:
            raise NotImplementedError('_build recorded a topology other than lib/uresnet.py:22-123; '
                                      'only that hot path is implemented natively')
        specs, off = [], 0
        nd = len(self._dims) - 1
        for (name, kind, k, s, ci, co) in want:
            wshape = (k,) * nd + ((ci, co) if kind == 'conv' else (co, ci))
            n = int(np.prod(wshape))
            specs.append((name + '/weights', wshape, off, n)); off += n
            specs.append((name + '/BatchNorm/beta', (co,), off, co)); off += co
        if off != self._n_params:
            raise RuntimeError('parameter count mismatch: python %d vs native %d' % (off, self._n_params))
        self._specs = specs

    def _expected_layers(self):
        F = int(getattr(self, '_base_num_outputs', 16))
        ns = int(getattr(self, '_num_strides', 5))
        cin = int(self._dims[-1])
        L = []

        def unit(scope, ci, co, s):
            if not (ci == co and s == 1):
                L.append((scope + '/shortcut', 'conv', 1, s, ci, co))
            L.append((scope + '/resnet_conv1', 'conv', 3, s, ci, co))
            L.append((scope + '/resnet_conv2', 'conv', 3, 1, co, co))

        L.append(('UResNet/conv0', 'conv', 3, 1, cin, F))
        c = F
        for step in range(ns):
            unit('UResNet/resnet_module%d/module1' % step, c, 2 * c, 2)
            unit('UResNet/resnet_module%d/module2' % step, 2 * c, 2 * c, 1)
            c *= 2
        for step in range(ns):
            co = c // 2
            L.append(('UResNet/deconv%d' % step, 'deconv', 3, 2, c, co))
            unit('UResNet/resnet_module%d/module1' % (step + 5), c, co, 1)
            unit('UResNet/resnet_module%d/module2' % (step + 5), co, co, 1)
            c = co
        L.append(('UResNet/conv1', 'conv', 3, 1, c, F))
        L.append(('UResNet/conv2', 'conv', 3, 1, F, self._num_class))
        return L

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

    def _feed(self, x, cols, what):
        """numpy / torch input [N, cols] -> contiguous fp32 device tensor (the H2D copy completes on the
        current stream before the caller's buffer may be reused: torch copies from pageable memory
        synchronously)."""
        import torch
        if isinstance(x, torch.Tensor):
            t = x.to(device=self._device, dtype=torch.float32)
        else:
            t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(self._device)
        t = t.reshape(-1, cols).contiguous()
        return t

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
        doc = ['', 'loss', 'acc. all', 'acc. nonzero']
        if not fetch:
            return None, doc
        return [None, float(out[0]), float(out[1]), float(out[2])], doc

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

    def inference_labels(self, sess, input_data, as_numpy=True):
        """ana_step's shower/track label volume computed on the device (lib/ssnet_trainval.py:285-287);
        returns [N, *spatial] float32 instead of the full softmax."""
        import torch
        d = self._feed(input_data, self._data_size, 'data')
        n = int(d.shape[0])
        self._ensure_handle(n)
        out = torch.empty((n,) + tuple(int(x) for x in self._dims[:-1]), dtype=torch.float32, device=self._device)
        _lib.check(_lib.load().ursn_infer_labels(self._handle, self._ptr(d), n, self._ptr(out), self._stream(sess)))
        self._last_feed = {'input_data': d}
        return out.cpu().numpy() if as_numpy else out

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
        n = self._last_feed['input_data'].shape[0]
        total = int(n * vox.value * cs.value)
        valid = total - (cs.value - ch.value)  # a channel-slice view ends `ch` floats into its last voxel
        torch.cuda.synchronize(self._device)
        buf = (ctypes.c_float * total)()
        from . import hiprt
        hiprt.memcpy_d2h(buf, ptr.value, valid * 4)
        a = np.frombuffer(buf, dtype=np.float32).reshape(n, int(vox.value), cs.value)[:, :, :ch.value]
        return a.reshape((n,) + tuple(int(d) for d in self._level_dims(int(vox.value))) + (ch.value,)).copy()

    def _level_dims(self, voxels):
        sp = [int(d) for d in self._dims[:-1]]
        while int(np.prod(sp)) > voxels:
            sp = [d // 2 for d in sp]
        return sp
