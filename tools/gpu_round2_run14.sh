set -x
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -s -k "cfg3" > gpurun_out/r2_t14.log 2>&1; echo rc=$? >> gpurun_out/r2_t14.log; grep "V=\|passed\|failed\|rc=" gpurun_out/r2_t14.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_b14.json 2> gpurun_out/r2_b14.err; echo bench rc=$?
