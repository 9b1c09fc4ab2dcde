timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_api.py tests/test_gpu_fullsize.py -m gpu -x -q -k "lee or local" > gpurun_out/r4_ll_t.log 2>&1; tail -5 gpurun_out/r4_ll_t.log
timeout -k 10 200 python3 scripts/config_scale_probe.py local 2>/dev/null | cut -c1-400
