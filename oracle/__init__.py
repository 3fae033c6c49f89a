"""CPU oracle for the U-ResNet hot path -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference (DeepLearnPhysics/u-resnet) holds no tests, golden
vectors or fixtures, and its arithmetic lives in un-vendored TensorFlow 1.x
(`tensorflow.contrib.slim`), which is absent from this image together with a
Python 2 interpreter.  The oracle is therefore a CPU *restatement* of the
reference semantics, written twice independently:

* ``oracle.uresnet_np``    -- numpy, float64 by default, explicit loops over the
  filter taps and a hand-derived analytic backward pass;
* ``oracle.uresnet_torch`` -- torch-CPU functional ops + autograd (oneDNN convs),
  also the ``cpu_baseline`` ("port") leg of ``bench.py``.

The two are cross-checked against each other in ``tests/test_oracle.py`` and the
golden fixtures under ``tests/golden/`` are generated from them by
``tests/golden/make_golden.py``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  The product (``u-resnet_amd/``) never does.
"""
