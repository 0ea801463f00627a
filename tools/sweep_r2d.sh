#!/bin/bash
# GPU visit: packed-operand GEMM path: kernel tests, whole-model parity, bench packed vs not.
out=gpurun_out/${1:-r2d}
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_model.py tests/test_gpu_fullsize_parity.py tests/test_gpu_fullsize.py tests/test_gpu_cli.py tests/test_gpu_parallel.py -x -q --timeout 600 > $out/pytest.log 2>&1 </dev/null
tail -12 $out/pytest.log
run() {  # name, config, env...
  name=$1; cfg=$2; shift 2
  env "$@" timeout 300 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/bench_$name.log 2>&1 </dev/null
  echo "$name: $(tail -1 $out/bench_$name.log | python tools/bench_fields.py 2>&1 | tail -1)"
}
run c3_packed cfg3 A=1
run c3_unpacked cfg3 ASR_GEMM_PACKED=0
run c2_packed cfg2 A=1
run c2_unpacked cfg2 ASR_GEMM_PACKED=0

timeout 300 python bench.py --config cfg3 --steps 10 --warmup 3 --dropout 0 --no-cpu-baseline --no-extras > $out/bench_c3_nodrop.log 2>&1 </dev/null
echo "c3_nodrop: $(tail -1 $out/bench_c3_nodrop.log | python tools/bench_fields.py 2>&1 | tail -1)"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- \
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/prof_stdout.log 2>&1 </dev/null
tail -1 $out/prof_stdout.log | cut -c1-200
