mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_gemm.py -x -q -k "fp8" 2>&1 | tail -40 > gpurun_out/f8_tests.log
timeout 300 python tools/dev/f8_probe.py > gpurun_out/f8_bench.log 2>&1
cat gpurun_out/f8_tests.log; cat gpurun_out/f8_bench.log
