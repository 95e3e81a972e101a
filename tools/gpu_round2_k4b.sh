set -x
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x > gpurun_out/r2_k4b_t.log 2>&1 && \
tail -3 gpurun_out/r2_k4b_t.log && \
timeout -k 10 400 python bench.py --no-secondary > gpurun_out/r2_k4b_b.json 2> gpurun_out/r2_k4b_b.err; echo bench rc=$?
