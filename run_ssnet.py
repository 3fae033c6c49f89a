#!/usr/bin/env python
"""Python 3 driver with the reference's call sequence (run_ssnet.py:11-19):

    python run_ssnet.py config/train3d.cfg
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 run_ssnet.py config/train3d.cfg
"""
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import uresnet_amd  # noqa: E402,F401
from uresnet_amd import ssnet_trainval as api  # noqa: E402


def main(argv):
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:  # one process per GPU, RCCL over xGMI
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl")
    t = api.ssnet_trainval()
    for a in argv:
        if a.endswith('.cfg'): t.override_config(a)
    t.initialize()
    t.batch_process()
    t.reset()


if __name__ == '__main__':
    main(sys.argv)
