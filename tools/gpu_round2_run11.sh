set -x
python tools/k2_ablate.py tools/abl > gpurun_out/r2_k2_slp.txt 2>&1; cat gpurun_out/r2_k2_slp.txt | tail -3
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_b11.json 2> gpurun_out/r2_b11.err; echo bench rc=$?
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r2_t11.log 2>&1; echo rc=$? >> gpurun_out/r2_t11.log; tail -4 gpurun_out/r2_t11.log
