set -x
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r2_t13.log 2>&1; echo rc=$? >> gpurun_out/r2_t13.log; tail -4 gpurun_out/r2_t13.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r2_b13.json 2> gpurun_out/r2_b13.err; echo bench rc=$?
