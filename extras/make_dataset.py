#!/usr/bin/env python
"""``python -m extras.make_dataset --parser dummy --parser_params split [0.8,0.1]
--input_parser mfcc --output_file data.h5`` -- builds the HDF5 dataset (features on
the GPU)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_study_amd.cli import make_dataset_main as main  # noqa: E402

if __name__ == '__main__':
    main()
