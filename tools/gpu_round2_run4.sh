set -x
python -m pytest tests -m gpu -q > gpurun_out/r2_t4.log 2>&1; echo rc=$? >> gpurun_out/r2_t4.log; tail -8 gpurun_out/r2_t4.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r2_b4.json 2> gpurun_out/r2_b4.err; echo bench rc=$?
