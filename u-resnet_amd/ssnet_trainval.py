"""Python 3 mirror of the reference orchestrator lib/ssnet_trainval.py (class ssnet_trainval).

Same public methods and the same call sequence into the network object:
``override_config* -> initialize -> batch_process (train_step | ana_step) -> reset``
(run_ssnet.py:11-19).  Differences, all outside the hot path:

* input comes from ``synthetic_io.synthetic_threadio`` (larcv2 / ROOT are not available); the
  MAIN_INPUT_CONFIG file is a synthetic-source description (config/input_synth_*.cfg);
* ``tf.Session`` -> ``HipSession``; TensorBoard summaries -> one JSON line per summary step under LOGDIR;
* ``tf.train.Saver`` -> ``.npz`` snapshots keyed by the TF variable names (SAVE_FILE-<iteration>.npz), resume
  parses the iteration from the file name like the reference (lib/ssnet_trainval.py:41-42,139-151);
* data parallelism (absent in the reference): under torch.distributed every rank reads its own entries,
  gradients are summed by ``apply_gradients`` and the reported metrics are averaged over ranks.
"""
from __future__ import print_function

import datetime
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


class ssnet_trainval(object):

    def __init__(self):
        self._cfg = ssnet_config()
        self._input_main = None
        self._input_test = None
        self._output = None
        self._iteration = -1
        self._sess = None

    def __del__(self):
        try:
            self.reset()
        except Exception:
            pass

    def _report(self, metrics, descr):
        msg = ''
        for i, desc in enumerate(descr):
            if not desc: continue
            msg += '%s=%6.6f   ' % (desc, metrics[i])
        msg += '\n'
        sys.stdout.write(msg)
        sys.stdout.flush()

    def num_class(self):
        return self._cfg.NUM_CLASS

    def iteration_from_file_name(self, file_name):
        name = file_name[:-4] if file_name.endswith('.npz') else file_name
        return int((name.split('-'))[-1])

    def override_config(self, file_name):
        self._cfg.override(file_name)
        self._cfg.dump()

    def report_memory(self):
        import torch
        return float(torch.cuda.max_memory_allocated())

    def _make_io(self, name, cfg_file, batch):
        io = synthetic_threadio()
        io.configure({'filler_name': name, 'verbosity': 0, 'filler_cfg': cfg_file})
        d = _dist()
        if d is not None and d.get_world_size() > 1:  # rank r reads entries r, r+W, r+2W, ...
            io._offset, io._stride = d.get_rank(), d.get_world_size()
        io.start_manager(batch)
        return io

    def initialize(self):
        if not self._cfg.MAIN_INPUT_CONFIG:
            print('Must provide larcv data filler configuration file!')
            return
        self._input_main = self._make_io('MainIO', self._cfg.MAIN_INPUT_CONFIG, self._cfg.MINIBATCH_SIZE)
        if self._cfg.TEST_INPUT_CONFIG:
            self._input_test = self._make_io('TestIO', self._cfg.TEST_INPUT_CONFIG, self._cfg.TEST_BATCH_SIZE)
        if self._cfg.ANA_OUTPUT_CONFIG:
            self._output = open(self._cfg.ANA_OUTPUT_CONFIG, 'ab')  # appended ssnet label volumes (.npy records)

        # image dimensions come from the data, not the cfg (lib/ssnet_trainval.py:89-93)
        self._input_main.next(store_entries=(not self._cfg.TRAIN), store_event_ids=(not self._cfg.TRAIN))
        dim_data = self._input_main.fetch_data(self._cfg.KEYWORD_DATA).dim()
        self._net = uresnet(dims=dim_data[1:], num_class=self._cfg.NUM_CLASS,
                            base_num_outputs=self._cfg.BASE_NUM_FILTERS, debug=self._cfg.DEBUG)
        if self._cfg.TRAIN:
            self._net.construct(trainable=self._cfg.TRAIN, use_weight=self._cfg.USE_WEIGHTS,
                                learning_rate=self._cfg.LEARNING_RATE, seed=self._cfg.TF_RANDOM_SEED)
        else:
            self._net.construct(trainable=self._cfg.TRAIN, use_weight=self._cfg.USE_WEIGHTS,
                                seed=self._cfg.TF_RANDOM_SEED)
        self._sess = HipSession()
        self._log_train = self._log_test = None
        rank0 = _dist() is None or _dist().get_rank() == 0
        if self._cfg.LOGDIR and rank0:
            for sub in ['train'] + (['test'] if self._input_test else []):
                os.makedirs(os.path.join(self._cfg.LOGDIR, sub), exist_ok=True)
            self._log_train = open(os.path.join(self._cfg.LOGDIR, 'train', 'scalars.jsonl'), 'a')
            if self._input_test:
                self._log_test = open(os.path.join(self._cfg.LOGDIR, 'test', 'scalars.jsonl'), 'a')
        if self._cfg.SAVE_FILE and rank0:
            save_dir = self._cfg.SAVE_FILE[0:self._cfg.SAVE_FILE.rfind('/')] if '/' in self._cfg.SAVE_FILE else ''
            if save_dir and not os.path.isdir(save_dir):
                os.makedirs(save_dir)
        if self._cfg.LOAD_FILE:
            self._iteration = self.iteration_from_file_name(self._cfg.LOAD_FILE)
            path = self._cfg.LOAD_FILE if self._cfg.LOAD_FILE.endswith('.npz') else self._cfg.LOAD_FILE + '.npz'
            with np.load(path, allow_pickle=False) as f:
                values = {}
                for name in self._net.variable_names():
                    if name in self._cfg.AVOID_LOAD_PARAMS or (name + ':0') in self._cfg.AVOID_LOAD_PARAMS:
                        print('\033[91mSkipping\033[00m loading variable', name, 'from input weight...')
                        continue
                    print('\033[95mLoading\033[00m variable', name, 'from', self._cfg.LOAD_FILE)
                    values[name] = f[name]
            self._net.set_variables(values, strict=False)
        self._batch_metrics = None
        self._descr_metrics = None
        self._saved = []

    def _fetch(self, io, kd, kl, kw):
        data = io.fetch_data(kd).data()
        label = io.fetch_data(kl).data()
        weight = None
        if self._cfg.USE_WEIGHTS:
            weight = io.fetch_data(kw).data()
            # perform per-event normalization, in place on the IO buffer (lib/ssnet_trainval.py:173)
            weight /= (np.sum(weight, axis=1).reshape([weight.shape[0], 1]))
        return data, label, weight

    def train_step(self):
        self._iteration += 1
        report_step = self._iteration % self._cfg.REPORT_STEPS == 0
        summary_step = self._cfg.SUMMARY_STEPS and (self._iteration % self._cfg.SUMMARY_STEPS) == 0
        checkpt_step = self._cfg.CHECKPOINT_STEPS and ((self._iteration + 1) % self._cfg.CHECKPOINT_STEPS) == 0

        self._net.zero_gradients(self._sess)
        for j in range(self._cfg.NUM_MINIBATCHES):
            minibatch_data, minibatch_label, minibatch_weight = self._fetch(
                self._input_main, self._cfg.KEYWORD_DATA, self._cfg.KEYWORD_LABEL, self._cfg.KEYWORD_WEIGHT)
            res, doc = self._net.accum_gradients(sess=self._sess, input_data=minibatch_data,
                                                 input_label=minibatch_label, input_weight=minibatch_weight)
            if self._batch_metrics is None:
                self._batch_metrics = np.zeros((self._cfg.NUM_MINIBATCHES, len(res) - 1), dtype=np.float32)
                self._descr_metrics = doc[1:]
            self._batch_metrics[j, :] = res[1:]
            self._input_main.next(store_entries=(not self._cfg.TRAIN), store_event_ids=(not self._cfg.TRAIN))
        self._net.apply_gradients(self._sess)  # all-reduce(sum) over ranks + Adam

        test_data = test_label = test_weight = None
        if (report_step or summary_step) and self._input_test:
            self._input_test.next()
            test_data, test_label, test_weight = self._fetch(
                self._input_test, self._cfg.KEYWORD_TEST_DATA, self._cfg.KEYWORD_TEST_LABEL,
                self._cfg.KEYWORD_TEST_WEIGHT)

        train_mean = self._mean_over_ranks(np.mean(self._batch_metrics, axis=0))
        rank0 = _dist() is None or _dist().get_rank() == 0
        if report_step:
            res = doc = None
            if self._input_test:
                res, doc = self._net.run_test(self._sess, test_data, test_label, test_weight)
            if rank0:
                tstamp = datetime.datetime.fromtimestamp(time.time()).strftime('%Y-%m-%d %H:%M:%S')
                sys.stdout.write('@ iteration {:d} LR {:g} Mem {:g} @ {:s}\n'.format(
                    self._iteration, self._net._opt._lr, self.report_memory(), tstamp))
                sys.stdout.write('Train set: ')
                self._report(train_mean, self._descr_metrics)
                if res is not None:
                    sys.stdout.write('Test set: ')
                    self._report(res, doc)
        if summary_step:
            summ = self._net.make_summary(self._sess, minibatch_data, minibatch_label, minibatch_weight)
            if self._log_train:
                summ['iteration'] = self._iteration
                self._log_train.write(json.dumps(summ) + '\n'); self._log_train.flush()
            if self._log_test and test_data is not None:
                summ = self._net.make_summary(self._sess, test_data, test_label, test_weight)
                summ['iteration'] = self._iteration
                self._log_test.write(json.dumps(summ) + '\n'); self._log_test.flush()
        if checkpt_step and self._cfg.SAVE_FILE and rank0:
            path = self.save_checkpoint()
            print('saved @', path)

    def _mean_over_ranks(self, metrics):
        d = _dist()
        if d is None or d.get_world_size() == 1:
            return metrics
        import torch
        t = torch.tensor(np.asarray(metrics, np.float64), device=self._net._device)
        d.all_reduce(t, op=d.ReduceOp.SUM)
        return (t / d.get_world_size()).cpu().numpy()

    def save_checkpoint(self):
        """SAVE_FILE-<iteration>.npz keyed by TF variable names; keeps CHECKPOINT_NMAX files."""
        path = '%s-%d.npz' % (self._cfg.SAVE_FILE, self._iteration)
        np.savez(path, **self._net.get_variables())
        self._saved.append(path)
        while len(self._saved) > max(int(self._cfg.CHECKPOINT_NMAX), 1):
            old = self._saved.pop(0)
            if os.path.isfile(old):
                os.remove(old)
        return path

    def ana(self, input_data, input_label=None):
        return self._net.inference(sess=self._sess, input_data=input_data, input_label=input_label)

    def ana_step(self, batch_mode=False):
        self._iteration += 1
        batch_data = self._input_main.fetch_data(self._cfg.KEYWORD_DATA).data()
        batch_label = self._input_main.fetch_data(self._cfg.KEYWORD_LABEL).data()
        softmax, acc_all, acc_nonzero = self.ana(input_data=batch_data, input_label=batch_label)

        copy_data = copy_label = copy_entries = None
        if not batch_mode:
            img_shape = list(softmax.shape)
            img_shape[-1] = -1
            copy_data = np.array(batch_data).reshape(img_shape)
            copy_label = np.array(batch_label).reshape(img_shape)
            copy_entries = np.array(self._input_main.fetch_entries())

        if self._output:
            entries = self._input_main.fetch_entries()
            for entry in range(len(softmax)):
                print('Entry', entries[entry], 'Acc', acc_nonzero)
                data = np.array(batch_data[entry]).reshape(softmax.shape[1:-1])
                shower_score = softmax[entry, ..., 1]
                track_score = softmax[entry, ..., 2]
                # lib/ssnet_trainval.py:285-287
                ssnet_result = (shower_score > track_score).astype(np.float32) + \
                    (track_score >= shower_score).astype(np.float32) * 2.0
                nonzero_map = (data > 1.0).astype(np.int32)
                ssnet_result = (ssnet_result * nonzero_map).astype(np.float32)
                np.save(self._output, ssnet_result)

        self._input_main.next(store_entries=(not self._cfg.TRAIN), store_event_ids=(not self._cfg.TRAIN))
        if not batch_mode:
            return {'entries': copy_entries, 'input': copy_data, 'label': copy_label, 'softmax': softmax,
                    'acc_all': acc_all, 'acc_nonzero': acc_nonzero}

    def batch_process(self):
        for i in range(self._cfg.ITERATIONS):
            if self._cfg.TRAIN and self._iteration >= self._cfg.ITERATIONS:
                print('Finished training (iteration %d)' % self._iteration)
                break
            if self._cfg.TRAIN:
                self.train_step()
            else:
                self.ana_step(batch_mode=True)

    def iterations(self):
        return self._cfg.ITERATIONS

    def current_iteration(self):
        return self._iteration

    def reset(self):
        if getattr(self, '_input_main', None) is not None:
            self._input_main.reset()
            self._input_main = None
        if getattr(self, '_input_test', None) is not None:
            self._input_test.reset()
            self._input_test = None
        if getattr(self, '_output', None) is not None:
            self._output.close()
            self._output = None
        for f in ('_log_train', '_log_test'):
            if getattr(self, f, None) is not None:
                getattr(self, f).close()
                setattr(self, f, None)
