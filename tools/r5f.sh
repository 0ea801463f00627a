mkdir -p gpurun_out/r5f
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r5f/test_all.log 2>&1; echo "rc=$?" >> gpurun_out/r5f/test_all.log
tail -25 gpurun_out/r5f/test_all.log
