#!/bin/bash
# GPU visit: packed GEMM kernels (128 and 256 tiles), RNG streams, whole-model parity on the
# packed path, microbenchmark of the layer's GEMM shapes, bench packed / per-tile.
out=gpurun_out/${1:-r2e}
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_random.py -q --timeout 600 > $out/pytest_a.log 2>&1 </dev/null
tail -6 $out/pytest_a.log
ASR_GEMM_HL_TILE=128 timeout 600 python -m pytest tests/test_gpu_gemm.py -q -k "hl" --timeout 600 > $out/pytest_a128.log 2>&1 </dev/null
tail -3 $out/pytest_a128.log
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize_parity.py tests/test_gpu_fullsize.py tests/test_gpu_cli.py tests/test_gpu_parallel.py -x -q --timeout 600 > $out/pytest_b.log 2>&1 </dev/null
tail -6 $out/pytest_b.log
timeout 300 python tools/gemm_hl_microbench.py > $out/mb256.log 2>&1 </dev/null; cat $out/mb256.log | tail -12
ASR_GEMM_HL_TILE=128 timeout 300 python tools/gemm_hl_microbench.py > $out/mb128.log 2>&1 </dev/null; grep packed $out/mb128.log
run() {  # name, config, env...
  name=$1; cfg=$2; shift 2
  env "$@" timeout 300 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/bench_$name.log 2>&1 </dev/null
  echo "$name: $(tail -1 $out/bench_$name.log | python tools/bench_fields.py 2>&1 | tail -1)"
}
run c3_packed256 cfg3 A=1
run c3_unpacked cfg3 ASR_GEMM_PACKED=0
run c2_packed256 cfg2 A=1
run c2_unpacked cfg2 ASR_GEMM_PACKED=0
