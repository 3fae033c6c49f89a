"""Python 3 driver with the public surface of the reference orchestrator lib/ssnet_trainval.py
(class ssnet_trainval): ``override_config* -> initialize -> batch_process (train_step | ana_step)
-> reset`` (run_ssnet.py:11-19), the same calls into the network object, the same stdout report and
result dictionary.  The body is organised around what this implementation has to get right that
TensorFlow hid: the device never waits for the host.

* a training iteration is ``_IterationPlan`` (which of report / summary / checkpoint fire,
  lib/ssnet_trainval.py:158-161) + ``_run_minibatches`` (zero -> accumulate x NUM_MINIBATCHES ->
  all-reduce + Adam, lib/ssnet_trainval.py:164-191).  Minibatches are launched with ``fetch=False``:
  the H2D copy of minibatch k+1 (pinned buffers, copy stream: ssnet.py::_feed) overlaps the kernels
  of minibatch k, and the three metric scalars are only read back on iterations that report them;
* the ana path writes the shower/track label volume computed on the device
  (lib/ssnet_trainval.py:285-287 -> ursn_infer_labels); the full softmax only crosses PCIe when the
  caller asked for it (``batch_mode=False`` returns it, as the reference does);
* input comes from ``synthetic_io.synthetic_threadio`` (larcv2 / ROOT are not available); ``tf.Session``
  -> ``HipSession``; TensorBoard summaries -> one JSON line per summary step under LOGDIR;
  ``tf.train.Saver`` -> ``.npz`` snapshots keyed by the TF variable names (SAVE_FILE-<iteration>.npz),
  resume parses the iteration from the file name like the reference (lib/ssnet_trainval.py:41-42,139-151);
* data parallelism (absent in the reference): under torch.distributed every rank reads its own entries,
  gradients are summed inside ``apply_gradients`` and reported metrics are averaged over ranks; only
  rank 0 prints, logs and saves.
"""
from __future__ import print_function

import collections
import datetime
import glob
import json
import os
import sys
import time

import numpy as np

from .config import ssnet_config
from .ssnet import HipSession
from .synthetic_io import synthetic_threadio
from .uresnet import uresnet


def _dist():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist
    except Exception:
        pass
    return None


def _is_rank0():
    d = _dist()
    return d is None or d.get_rank() == 0


_IterationPlan = collections.namedtuple('_IterationPlan', 'iteration report summary checkpoint')


class _ScalarLog(object):
    """Replaces tf.summary.FileWriter (lib/ssnet_trainval.py:118-127): JSON lines under LOGDIR/<sub>/."""

    def __init__(self, logdir, sub):
        os.makedirs(os.path.join(logdir, sub), exist_ok=True)
        self._f = open(os.path.join(logdir, sub, 'scalars.jsonl'), 'a')

    def add_summary(self, summary, iteration):
        rec = dict(summary)
        rec['iteration'] = iteration
        self._f.write(json.dumps(rec) + '\n')
        self._f.flush()

    def close(self):
        self._f.close()


class ssnet_trainval(object):

    def __init__(self):
        self._cfg = ssnet_config()
        self._input_main = None
        self._input_test = None
        self._output = None
        self._iteration = -1
        self._sess = None
        self._net = None
        self._writer_train = self._writer_test = None

    def __del__(self):
        try:
            self.reset()
        except Exception:
            pass

    # ---- small public helpers of the reference -----------------------------------------------------------
    def _report(self, metrics, descr):
        """lib/ssnet_trainval.py:29-36: 'name=value   ' per documented metric."""
        sys.stdout.write(''.join('%s=%6.6f   ' % (d, m) for m, d in zip(metrics, descr) if d) + '\n')
        sys.stdout.flush()

    def num_class(self):
        return self._cfg.NUM_CLASS

    def iteration_from_file_name(self, file_name):
        stem = file_name[:-4] if file_name.endswith('.npz') else file_name
        return int(stem.rsplit('-', 1)[-1])

    def override_config(self, file_name):
        self._cfg.override(file_name)
        self._cfg.dump()

    def report_memory(self):
        import torch
        return float(torch.cuda.max_memory_allocated())

    def iterations(self):
        return self._cfg.ITERATIONS

    def current_iteration(self):
        return self._iteration

    # ---- initialize (lib/ssnet_trainval.py:51-154) ---------------------------------------------------------
    def _open_stream(self, name, cfg_file, batch):
        io = synthetic_threadio()
        io.configure({'filler_name': name, 'verbosity': 0, 'filler_cfg': cfg_file})
        d = _dist()
        if d is not None and d.get_world_size() > 1:
            io.shard(d.get_rank(), d.get_world_size())   # rank r reads entries r, r+W, r+2W, ...
        io.start_manager(batch)
        return io

    def _advance_main(self):
        keep = not self._cfg.TRAIN
        self._input_main.next(store_entries=keep, store_event_ids=keep)

    def initialize(self):
        cfg = self._cfg
        if not cfg.MAIN_INPUT_CONFIG:
            print('Must provide larcv data filler configuration file!')
            return
        self._input_main = self._open_stream('MainIO', cfg.MAIN_INPUT_CONFIG, cfg.MINIBATCH_SIZE)
        if cfg.TEST_INPUT_CONFIG:
            self._input_test = self._open_stream('TestIO', cfg.TEST_INPUT_CONFIG, cfg.TEST_BATCH_SIZE)
        if cfg.ANA_OUTPUT_CONFIG:
            self._output = open(cfg.ANA_OUTPUT_CONFIG, 'ab')   # appended ssnet label volumes (.npy records)

        # image dimensions come from the first batch, not from the cfg (lib/ssnet_trainval.py:89-93)
        self._advance_main()
        dims = self._input_main.fetch_data(cfg.KEYWORD_DATA).dim()[1:]
        self._net = uresnet(dims=dims, num_class=cfg.NUM_CLASS, base_num_outputs=cfg.BASE_NUM_FILTERS, debug=cfg.DEBUG)
        extra = {'learning_rate': cfg.LEARNING_RATE} if cfg.TRAIN else {}
        self._net.construct(trainable=cfg.TRAIN, use_weight=cfg.USE_WEIGHTS, seed=cfg.TF_RANDOM_SEED,
                            precision=cfg.PRECISION, **extra)
        self._sess = HipSession()

        self._saved = collections.deque()
        if _is_rank0():
            if cfg.LOGDIR:
                self._writer_train = _ScalarLog(cfg.LOGDIR, 'train')
                if self._input_test:
                    self._writer_test = _ScalarLog(cfg.LOGDIR, 'test')
            if cfg.SAVE_FILE:
                save_dir = os.path.dirname(cfg.SAVE_FILE)
                if save_dir:
                    os.makedirs(save_dir, exist_ok=True)
                # snapshots written before a restart take part in the CHECKPOINT_NMAX rotation
                old = glob.glob(glob.escape(cfg.SAVE_FILE) + '-*.npz')
                for path in sorted((p for p in old if p[len(cfg.SAVE_FILE) + 1:-4].isdigit()),
                                   key=self.iteration_from_file_name):
                    self._saved.append(path)
        if cfg.LOAD_FILE:
            self._restore(cfg.LOAD_FILE)
        self._descr_metrics = None

    def _restore(self, load_file):
        self._iteration = self.iteration_from_file_name(load_file)
        path = load_file if load_file.endswith('.npz') else load_file + '.npz'
        skip = set(self._cfg.AVOID_LOAD_PARAMS)
        values = {}
        with np.load(path, allow_pickle=False) as f:
            for name in self._net.variable_names():
                if name in skip or (name + ':0') in skip:
                    print('\033[91mSkipping\033[00m loading variable', name, 'from input weight...')
                    continue
                print('\033[95mLoading\033[00m variable', name, 'from', load_file)
                values[name] = f[name]
        self._net.set_variables(values, strict=False)

    # ---- training (lib/ssnet_trainval.py:156-233) ----------------------------------------------------------
    def _plan_iteration(self):
        self._iteration += 1
        it, c = self._iteration, self._cfg
        return _IterationPlan(iteration=it,
                              report=it % c.REPORT_STEPS == 0,
                              summary=bool(c.SUMMARY_STEPS) and it % c.SUMMARY_STEPS == 0,
                              checkpoint=bool(c.CHECKPOINT_STEPS) and (it + 1) % c.CHECKPOINT_STEPS == 0)

    def _pull(self, io, kd, kl, kw):
        """One batch off an IO stream; the per-event weight normalisation mutates the IO buffer in place
        (lib/ssnet_trainval.py:173)."""
        data = io.fetch_data(kd).data()
        label = io.fetch_data(kl).data()
        weight = None
        if self._cfg.USE_WEIGHTS:
            weight = io.fetch_data(kw).data()
            weight /= np.sum(weight, axis=1).reshape([weight.shape[0], 1])
        return data, label, weight

    def _run_minibatches(self, want_metrics):
        """zero -> NUM_MINIBATCHES x accumulate -> apply.  Returns the per-minibatch metrics [M, 3] when asked for
        (each read is a stream synchronisation), else None; in both cases ``self._last_minibatch`` holds the
        device-resident tensors of the last minibatch for a summary."""
        c, net = self._cfg, self._net
        rows = []
        net.zero_gradients(self._sess)
        for _ in range(c.NUM_MINIBATCHES):
            data, label, weight = self._pull(self._input_main, c.KEYWORD_DATA, c.KEYWORD_LABEL, c.KEYWORD_WEIGHT)
            res, doc = net.accum_gradients(sess=self._sess, input_data=data, input_label=label, input_weight=weight,
                                           fetch=want_metrics)
            # the copy has completed, the kernels are queued: the IO may refill this buffer while they run
            self._descr_metrics = doc[1:]
            if want_metrics:
                rows.append(res[1:])
            self._advance_main()
        self._last_minibatch = net.last_feed()
        net.apply_gradients(self._sess)   # all-reduce(sum) over ranks + Adam
        return np.asarray(rows, np.float32) if want_metrics else None

    def _mean_over_ranks(self, metrics):
        d = _dist()
        if d is None or d.get_world_size() == 1:
            return metrics
        import torch
        dev = self._net._device if d.get_backend() == 'nccl' else 'cpu'
        t = torch.tensor(np.asarray(metrics, np.float64), device=dev)
        d.all_reduce(t, op=d.ReduceOp.SUM)
        return (t / d.get_world_size()).cpu().numpy()

    def train_step(self):
        plan = self._plan_iteration()
        per_minibatch = self._run_minibatches(want_metrics=plan.report)

        test = None
        if (plan.report or plan.summary) and self._input_test:
            c = self._cfg
            self._input_test.next()
            test = self._pull(self._input_test, c.KEYWORD_TEST_DATA, c.KEYWORD_TEST_LABEL, c.KEYWORD_TEST_WEIGHT)

        if plan.report:
            # report_step is the same on every rank, so the collective inside is safe
            train_mean = self._mean_over_ranks(per_minibatch.mean(axis=0))
            tested = self._net.run_test(self._sess, *test) if test is not None else None
            if _is_rank0():
                stamp = datetime.datetime.fromtimestamp(time.time()).strftime('%Y-%m-%d %H:%M:%S')
                sys.stdout.write('@ iteration {:d} LR {:g} Mem {:g} @ {:s}\n'.format(
                    plan.iteration, self._net._opt._lr, self.report_memory(), stamp))
                sys.stdout.write('Train set: ')
                self._report(train_mean, self._descr_metrics)
                if tested is not None:
                    sys.stdout.write('Test set: ')
                    self._report(*tested)
        if plan.summary:
            last = self._last_minibatch
            summ = self._net.make_summary(self._sess, last['input_data'], last['input_label'], last.get('input_weight'))
            if self._writer_train:
                self._writer_train.add_summary(summ, plan.iteration)
            if self._writer_test and test is not None:
                self._writer_test.add_summary(self._net.make_summary(self._sess, *test), plan.iteration)
        if plan.checkpoint and self._cfg.SAVE_FILE and _is_rank0():
            print('saved @', self.save_checkpoint())

    def save_checkpoint(self):
        """SAVE_FILE-<iteration>.npz keyed by TF variable names; keeps the newest CHECKPOINT_NMAX files."""
        path = '%s-%d.npz' % (self._cfg.SAVE_FILE, self._iteration)
        np.savez(path, **self._net.get_variables())
        if path in self._saved:
            self._saved.remove(path)
        self._saved.append(path)
        while len(self._saved) > max(int(self._cfg.CHECKPOINT_NMAX), 1):
            old = self._saved.popleft()
            if os.path.isfile(old):
                os.remove(old)
        return path

    # ---- analysis (lib/ssnet_trainval.py:235-314) ----------------------------------------------------------
    def ana(self, input_data, input_label=None):
        return self._net.inference(sess=self._sess, input_data=input_data, input_label=input_label)

    def ana_step(self, batch_mode=False):
        self._iteration += 1
        c, io = self._cfg, self._input_main
        batch_data = io.fetch_data(c.KEYWORD_DATA).data()
        batch_label = io.fetch_data(c.KEYWORD_LABEL).data()
        entries = io.fetch_entries()

        softmax = labels = None
        if self._output:
            # the label rule runs inside the head kernel; the softmax is only brought back when it is returned
            out = self._net.inference_labels(self._sess, batch_data, batch_label, with_softmax=not batch_mode)
            labels, acc_all, acc_nonzero = out[0], out[1], out[2]
            if not batch_mode:
                softmax = out[3]
            for i in range(labels.shape[0]):
                print('Entry', entries[i], 'Acc', acc_nonzero)
                np.save(self._output, labels[i])
            self._output.flush()
        else:
            softmax, acc_all, acc_nonzero = self.ana(input_data=batch_data, input_label=batch_label)

        result = None
        if not batch_mode:
            img_shape = list(softmax.shape)
            img_shape[-1] = -1
            result = {'entries': np.array(entries), 'input': np.array(batch_data).reshape(img_shape),
                      'label': np.array(batch_label).reshape(img_shape), 'softmax': softmax,
                      'acc_all': acc_all, 'acc_nonzero': acc_nonzero}
        self._advance_main()
        return result

    def batch_process(self):
        for _ in range(self._cfg.ITERATIONS):
            if self._cfg.TRAIN and self._iteration >= self._cfg.ITERATIONS:
                print('Finished training (iteration %d)' % self._iteration)
                break
            if self._cfg.TRAIN:
                self.train_step()
            else:
                self.ana_step(batch_mode=True)

    def reset(self):
        for attr in ('_input_main', '_input_test'):
            io = getattr(self, attr, None)
            if io is not None:
                io.reset()
                setattr(self, attr, None)
        for attr in ('_output', '_writer_train', '_writer_test'):
            f = getattr(self, attr, None)
            if f is not None:
                f.close()
                setattr(self, attr, None)
