"""Synthetic stand-in for larcv's threaded batch filler (``larcv.dataloader2.larcv_threadio``).

The reference reads LArTPC images/volumes from ROOT files through larcv2 (lib/ssnet_trainval.py:64-91,
167-188; config/input_train3d.cfg); neither larcv2 nor the data files exist here, so the driver is fed
by this class, which keeps the call protocol the reference uses:

    io = synthetic_threadio(); io.configure(cfg_dict); io.start_manager(batch_size)
    io.next(store_entries=..., store_event_ids=...)
    io.fetch_data(key).dim()   -> [N, *dims]         io.fetch_data(key).data() -> float32 [N, prod(dims)]
    io.fetch_entries(); io.fetch_event_ids(); io.reset()

Batches are produced by a background thread one step ahead (like larcv's filler threads), into two
alternating sets of PAGE-LOCKED host buffers when a HIP device is present (the network's feed then copies
straight out of them on its copy stream, ssnet.py::_feed; a buffer is refilled only after the ``next()`` that
follows its use, which is the validity contract of larcv's buffers, lib/ssnet_trainval.py:167-188), from two
generators: ``dense_uniform`` (random pixels) and ``lartpc_sparse`` (a few straight
"tracks" and blob "showers" on an empty background, values 1..500, inverse-frequency weights).
Samples are seeded ``1234 + entry`` (1234 = TF_RANDOM_SEED, lib/config.py:22) so any rank/any run
can regenerate entry k.
"""
from __future__ import print_function

import ast
import threading

import numpy as np


def dense_uniform(dims, num_class, entry):
    rng = np.random.default_rng(1234 + int(entry))
    lsz = int(np.prod(dims[:-1]))
    data = rng.random(int(np.prod(dims)), dtype=np.float32)
    label = rng.integers(0, num_class, lsz).astype(np.float32)
    weight = np.ones(lsz, np.float32)
    return data, label, weight


def lartpc_sparse(dims, num_class, entry):
    """Sparse track/shower toy event.  label 0 background, 1 shower, 2 track (3-class); with more
    classes the extra labels 3.. are assigned per segment."""
    assert int(dims[-1]) == 1
    sp = tuple(int(d) for d in dims[:-1])
    nd = len(sp)
    rng = np.random.default_rng(1234 + int(entry))
    data = np.zeros(sp, np.float32)
    label = np.zeros(sp, np.float32)
    size = np.array(sp, np.float64)

    def deposit(pts, lab):
        idx = np.round(pts).astype(np.int64)
        ok = np.all((idx >= 0) & (idx < np.array(sp)), axis=1)
        idx = idx[ok]
        if idx.size == 0:
            return
        val = np.clip(1.0 + rng.exponential(20.0, idx.shape[0]), 1.0, 500.0).astype(np.float32)
        tup = tuple(idx[:, j] for j in range(nd))
        data[tup] = val
        label[tup] = lab

    for k in range(int(rng.integers(1, 4))):  # showers: Gaussian blobs
        c = rng.uniform(0.2, 0.8, nd) * size
        sig = rng.uniform(0.02, 0.05) * size.min()
        npts = int(rng.integers(200, 800) * (size.min() / 64.0) ** (nd - 1))
        lab = 1 if num_class <= 3 else 1 + (k % 2) * 2
        deposit(c + rng.normal(0, sig, (npts, nd)), min(lab, num_class - 1))
    for k in range(int(rng.integers(2, 7))):  # tracks: straight segments
        a = rng.uniform(0.05, 0.95, nd) * size
        b = rng.uniform(0.05, 0.95, nd) * size
        npts = int(np.abs(b - a).max() * 2) + 2
        t = np.linspace(0, 1, npts)[:, None]
        lab = 2 if num_class <= 3 else 2 + (k % 2) * 2
        deposit(a + (b - a) * t, min(lab, num_class - 1))
    lab_i = label.astype(np.int64)
    counts = np.bincount(lab_i.ravel(), minlength=num_class).astype(np.float64)
    inv = np.where(counts > 0, 1.0 / np.maximum(counts, 1), 0.0)
    weight = inv[lab_i].astype(np.float32)
    return data.reshape(-1), label.reshape(-1), weight.reshape(-1)


GENERATORS = {'dense_uniform': dense_uniform, 'lartpc_sparse': lartpc_sparse}


class _batch_data(object):
    """What ``fetch_data(key)`` returns in larcv: ``.dim()`` and ``.data()``."""

    def __init__(self, arr, dim):
        self._arr, self._dim = arr, list(dim)

    def dim(self):
        return self._dim

    def data(self):
        return self._arr


def read_cfg(path):
    """``KEY VALUE`` lines (same format as the network cfg files): Dims, NumClass, Generator,
    NumEntries, Keys (dict: role -> keyword)."""
    out = {}
    with open(path) as f:
        for line in f.read().split('\n'):
            line = line.split('#')[0].strip()
            if not line:
                continue
            k, v = line.split(None, 1)
            out[k] = ast.literal_eval(v.strip())
    return out


class synthetic_threadio(object):

    def __init__(self):
        self._cfg = None
        self._thread = None
        self._ready = None
        self._cursor = 0
        self._batch = 0

    def configure(self, cfg):
        """cfg: dict with 'filler_cfg' = path of a synthetic input cfg or an inline dict
        (the reference passes {'filler_name','verbosity','filler_cfg'}, lib/ssnet_trainval.py:65-68)."""
        fc = cfg.get('filler_cfg', cfg)
        c = read_cfg(fc) if isinstance(fc, str) else dict(fc)
        self._dims = [int(d) for d in c['Dims']]
        self._num_class = int(c.get('NumClass', 3))
        self._gen = GENERATORS[c.get('Generator', 'lartpc_sparse')]
        self._num_entries = int(c.get('NumEntries', 1 << 30))
        self._keys = dict(c.get('Keys', {'data': 'data', 'label': 'label', 'weight': 'weight'}))
        self._offset = int(c.get('FirstEntry', 0))
        self._stride = int(c.get('EntryStride', 1))   # data parallel: rank r reads entries r, r+W, ...
        self._cfg = c

    def shard(self, rank, world_size):
        """Data parallelism: this reader serves entries rank, rank + W, rank + 2W, ... (call before start_manager)."""
        self._offset, self._stride = int(rank), int(world_size)

    def start_manager(self, batch_size):
        self._batch = int(batch_size)
        self._sets = [self._alloc_set(), self._alloc_set()]
        self._fill = 0
        self._spawn()

    @staticmethod
    def _host_buffer(shape):
        try:
            import torch
            if torch.cuda.is_available():
                return torch.empty(shape, dtype=torch.float32, pin_memory=True).numpy()
        except Exception:
            pass
        return np.empty(shape, np.float32)

    def _alloc_set(self):
        n = self._batch
        dsz, lsz = int(np.prod(self._dims)), int(np.prod(self._dims[:-1]))
        return dict(data=self._host_buffer((n, dsz)), label=self._host_buffer((n, lsz)),
                    weight=self._host_buffer((n, lsz)))

    def _make(self, first, bufs):
        n = self._batch
        data, label, weight = bufs['data'], bufs['label'], bufs['weight']
        entries = []
        for i in range(n):
            e = (self._offset + (first + i) * self._stride) % self._num_entries
            d, l, w = self._gen(self._dims, self._num_class, e)
            data[i], label[i], weight[i] = d, l, w
            entries.append(e)
        return dict(data=data, label=label, weight=weight, entries=entries)

    def _spawn(self):
        first = self._cursor
        self._cursor += self._batch
        box = {}
        bufs = self._sets[self._fill]   # the set published two next() calls ago: its batch is no longer valid
        self._fill ^= 1

        def work():
            box['b'] = self._make(first, bufs)
        self._thread = threading.Thread(target=work)
        self._thread.daemon = True
        self._thread.start()
        self._pending = box

    def next(self, store_entries=False, store_event_ids=False):
        """Publishes the prefetched batch and starts filling the following one."""
        self._thread.join()
        self._ready = self._pending['b']
        self._spawn()

    def fetch_data(self, key):
        if self._ready is None:
            self.next()
        role = None
        for r, k in self._keys.items():
            if k == key:
                role = r
        if role is None:
            raise KeyError('no producer named %r (have %r)' % (key, self._keys))
        n = self._batch
        dim = [n] + (self._dims if role == 'data' else self._dims[:-1])
        return _batch_data(self._ready[role], dim)

    def fetch_entries(self):
        return list(self._ready['entries'])

    def fetch_event_ids(self):
        return [(0, 0, e) for e in self._ready['entries']]

    def reset(self):
        if self._thread is not None:
            self._thread.join()
        self._thread = None
        self._ready = None
