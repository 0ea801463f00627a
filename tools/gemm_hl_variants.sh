cd /root/repo
export MB_ONLY=cfg3
for v in "$@"; do echo "== $v"; ASR_LIB_PATH=variants/libasr_$v.so timeout 120 python tools/gemm_hl_microbench.py 2>&1 | grep "fwd\|dX\|dW\|dU";  ASR_LIB_PATH=variants/libasr_$v.so timeout 120 python tools/gemm_hl_phase.py 2>&1 | grep "slabs\|wave 0\|wave 4"; done
