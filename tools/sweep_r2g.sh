#!/bin/bash
out=gpurun_out/${1:-r2g}
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_frontend.py tests/test_gpu_random.py tests/test_gpu_model.py -q --timeout 600 > $out/pytest.log 2>&1 </dev/null
tail -5 $out/pytest.log
bash tools/pmc_gemm_hl.sh $(basename $out)/pmc
