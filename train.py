#!/usr/bin/env python
"""``python train.py --dataset data.h5 --model brsmv1 ...`` -- the reference's training
command line on the MI355X hot path (flags: asr_study_amd/cli.py).  One process per
GPU for data parallelism:
``python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py ...``"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from asr_study_amd.cli import train_main as main  # noqa: E402

if __name__ == '__main__':
    main()
