set -x
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/r2_full_t.log 2>&1; echo rc=$? >> gpurun_out/r2_full_t.log; tail -4 gpurun_out/r2_full_t.log
