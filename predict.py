#!/usr/bin/env python
"""``python predict.py --model best.h5 --dataset data.h5 [--subset test]`` or
``--file utterance.wav`` -- the reference's transcription command line."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from asr_study_amd.cli import predict_main as main  # noqa: E402

if __name__ == '__main__':
    main()
