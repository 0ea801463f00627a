"""CPU oracle for the asr-study acoustic-training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / the reported CPU baseline.  The
product path (``asr_study_amd``) never imports this package and fails loudly
when the HIP library is missing.

Each module restates, in NumPy float64 (float32 where stated), the algorithm the
reference runs for one piece of the path and cites the reference file:line it
follows (paths relative to the reference checkout):

* ``frontend.py`` -- preprocessing/audio.py + preprocessing/audio_utils.py.
  PINNED: verified bit-for-bit (float64) against the reference's own NumPy code
  imported in the build container (``oracle/gen_golden.py`` -> tests/golden/).
* ``lstm.py``     -- core/layers.py:432-469 + Keras 1.2.2 LSTM/Bidirectional/
  TimeDistributed(Dense) semantics.  PARITY UNPINNED by the reference (Keras
  1.2.2 / TF 1.3.0 are un-vendored third-party wheels and cannot run here);
  cross-checked against torch.autograd in tests.
* ``ctc.py``      -- core/ctc_utils.py:60-70 -> tf.nn.ctc_loss (TF 1.3.0).
  PARITY UNPINNED by the reference; pinned instead on TensorFlow's two published
  known-answer vectors and torch.nn.functional.ctc_loss.
* ``decode.py``   -- core/ctc_utils.py:8-52, core/metrics.py:4-8 (greedy, prefix
  beam search, edit distance).  PARITY UNPINNED by the reference; brute force
  enumeration cross-checks in tests.
* ``optim.py``    -- train.py:133-137 (Keras Adam / SGD with global clipnorm).
"""
