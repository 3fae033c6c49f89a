"""Python 3 mirror of the reference flag system (lib/config.py:6-76).

Same class-attribute defaults (lib/config.py:8-36) and the same file format: one ``KEY VALUE``
per line, ``#`` starts a comment, VALUE is a Python literal.  Differences, chosen on purpose
(SURVEY.md Appendix E): values are parsed with ``ast.literal_eval`` instead of ``exec``
(lib/config.py:61,66), and an unknown key is reported and skipped instead of crashing
(lib/config.py:57-61 prints the warning but then fails inside ``exec``), so the shipped
``config/ana3d.cfg`` loads.
"""
from __future__ import print_function

import ast
import os


class ssnet_config(object):

    NUM_CLASS = 3
    BASE_NUM_FILTERS = 16
    MAIN_INPUT_CONFIG = 'config/input_train.cfg'
    TEST_INPUT_CONFIG = ''
    ANA_OUTPUT_CONFIG = ''
    LOGDIR = 'ssnet_train_log'
    SAVE_FILE = 'ssnet_checkpoint/uresnet'
    LOAD_FILE = ''
    AVOID_LOAD_PARAMS = []
    LEARNING_RATE = -1
    MINIBATCH_SIZE = 10
    NUM_MINIBATCHES = 5
    TEST_BATCH_SIZE = 10
    ITERATIONS = 100000
    TF_RANDOM_SEED = 1234
    TRAIN = True
    DEBUG = False
    USE_WEIGHTS = True
    REPORT_STEPS = 200
    SUMMARY_STEPS = 20
    CHECKPOINT_STEPS = 200
    CHECKPOINT_NMAX = 10
    CHECKPOINT_NHOUR = 0.4
    KEYWORD_DATA = 'data'
    KEYWORD_LABEL = 'label'
    KEYWORD_WEIGHT = 'weight'
    KEYWORD_TEST_DATA = ''
    KEYWORD_TEST_LABEL = ''
    KEYWORD_TEST_WEIGHT = ''
    # not in the reference (fp32 TensorFlow): 'fp32' | 'bf16' -- bf16 = mixed precision of BASELINE.json configs[4]
    # (activations / gradient tensors bf16 in HBM, fp32 parameters, statistics, accumulators and Adam)
    PRECISION = 'fp32'

    def __init__(self):
        pass

    @classmethod
    def _keys(cls):
        return [s for s in ssnet_config.__dict__.keys() if s == s.upper() and not s.startswith('_')]

    def override(self, file_name):
        """lib/config.py:40-66.  Raises IOError for a missing file and TypeError for a value whose
        type differs from the default's (LEARNING_RATE excepted, lib/config.py:62)."""
        keys = self._keys()
        if not os.path.isfile(file_name):
            print('Config file not found', file_name)
            raise IOError(file_name)
        with open(file_name, 'r') as f:
            lines = f.read().split('\n')
        for line in lines:
            valid_line = line
            if line.find('#') >= 0:
                valid_line = line[0:line.find('#')]
            words = valid_line.split(None, 1)
            if len(words) == 0:
                continue
            if len(words) != 2:
                print('Ignoring a line:', line)
                continue
            key, text = words[0], words[1].strip()
            if key not in keys:
                print('Ignoring a parameter in file:', key)
                continue
            try:
                value = ast.literal_eval(text)
            except (ValueError, SyntaxError):
                print('Incompatible type: %s' % line)
                raise TypeError(line)
            if key != 'LEARNING_RATE' and type(getattr(self, key)) != type(value):
                print('Incompatible type: %s' % line)
                raise TypeError(line)
            if key == 'PRECISION' and value not in ('fp32', 'bf16'):
                print('Incompatible value: %s' % line)
                raise TypeError(line)
            setattr(self, key, value)

    def dump(self):
        """lib/config.py:68-76."""
        for key in self._keys():
            msg = key
            while len(msg) < 20:
                msg += '.'
            print('%s %s' % (msg, str(getattr(self, key))))


if __name__ == '__main__':
    import sys
    k = ssnet_config()
    k.dump()
    if len(sys.argv) > 1:
        k.override(sys.argv[1])
        print('\n')
        k.dump()
