cd /root/repo
timeout 600 python -m pytest tests/test_gpu_gemm.py -x -q -m gpu 2>&1 | tail -5
MB_ONLY=cfg3 timeout 120 python tools/gemm_hl_microbench.py 2>&1 | grep -v amdgpu.ids
timeout 120 python tools/gemm_hl_phase.py 2>&1 | grep "slabs\|wave 0\|wave 4"
