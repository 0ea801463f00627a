mkdir -p gpurun_out/r5j
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py -x -q -m gpu -k "deep_speech2 or compact or side_stream" > gpurun_out/r5j/test.log 2>&1; echo "rc=$?" >> gpurun_out/r5j/test.log
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/r5j/bench_full.log 2> gpurun_out/r5j/bench_full.err; echo rc=$?
tail -3 gpurun_out/r5j/test.log
