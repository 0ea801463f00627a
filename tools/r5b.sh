set -x
mkdir -p gpurun_out/r5b
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_random.py -x -q -m gpu -k "independent or toolchain" > gpurun_out/r5b/test1.log 2>&1; echo "rc=$?" >> gpurun_out/r5b/test1.log
timeout 900 python -m pytest tests/test_gpu_fullsize_parity.py -x -q -m gpu -s -k "cfg3_conv" > gpurun_out/r5b/test2.log 2>&1; echo "rc=$?" >> gpurun_out/r5b/test2.log
timeout 300 python tools/gemm_hl_phase.py > gpurun_out/r5b/phase.log 2>&1
timeout 300 python __graft_entry__.py smoke > gpurun_out/r5b/smoke.log 2>&1; echo "rc=$?" >> gpurun_out/r5b/smoke.log
tail -5 gpurun_out/r5b/test1.log; grep -E "parity|passed|failed|rc=" gpurun_out/r5b/test2.log | tail -30; cat gpurun_out/r5b/phase.log; tail -3 gpurun_out/r5b/smoke.log
