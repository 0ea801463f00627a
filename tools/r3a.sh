#!/bin/bash
# Round 3, first GPU visit: the new / changed -m gpu tests, then the cfg3 bench with the BPTT
# kernel in its one-dimensional (bwd_body_x) and two-dimensional (bwd_body_c) split.
tag=${1:-r3a}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest -m gpu -x -q --timeout 600 -s \
    tests/test_gpu_lstm.py::test_bptt_two_dimensional_split_matches_the_one_dimensional_kernel \
    tests/test_gpu_optim.py tests/test_gpu_parallel.py \
    "tests/test_gpu_gemm.py::test_pack_hl_planes_both_orientations" \
    "tests/test_gpu_model.py::test_brsmv1_packed_operand_gemm_path" \
    > $out/pytest_new.log 2>&1 </dev/null
tail -5 $out/pytest_new.log
for form in 0 1; do
  ASR_LSTM_BWD_2D=$form timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras \
      > $out/bench_2d$form.log 2>&1 </dev/null
  python tools/bench_fields.py $out/bench_2d$form.log
done
ASR_BENCH_CONFIG=cfg2 ASR_LSTM_BWD_2D=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras \
      > $out/bench_cfg2_2d1.log 2>&1 </dev/null
python tools/bench_fields.py $out/bench_cfg2_2d1.log
ASR_BENCH_CONFIG=cfg2 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras \
      > $out/bench_cfg2_2d0.log 2>&1 </dev/null
python tools/bench_fields.py $out/bench_cfg2_2d0.log
