#!/usr/bin/env python
"""``python eval.py --model best.h5 --dataset data.h5 [--subset test] [--beam_width 100]``
-- the reference's evaluation command line (CTC beam search + LER)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from asr_study_amd.cli import eval_main as main  # noqa: E402

if __name__ == '__main__':
    main()
