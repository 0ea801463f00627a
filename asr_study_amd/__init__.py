"""asr_study_amd -- MI355X-native acoustic-training hot path of igormq/asr-study.

MFCC / log-mel front-end -> bidirectional LSTM stack -> CTC loss / decode, with all
arithmetic in hand-written gfx950 HIP kernels behind the C ABI of
include/asr_hip.h (libasr_hip.so).  The sub-packages mirror the reference's layout
(preprocessing/, core/, datasets/, utils/) so `train.py` / `eval.py` read like the
reference's own.  torch is used only for device memory, streams and
torch.distributed (RCCL).
"""
__version__ = '0.1.0'
