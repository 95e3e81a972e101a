set -x
LCAO_K2_PIPE=0 python tools/k2_ablate.py tools/abl > gpurun_out/r2_k2_pipe.txt 2>&1
LCAO_K2_PIPE=1 python tools/k2_ablate.py tools/abl >> gpurun_out/r2_k2_pipe.txt 2>&1
grep lib_ gpurun_out/r2_k2_pipe.txt
LCAO_K2_PIPE=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fused_mfma or cfg1 or cfg2 or golden or edge_order" > gpurun_out/r2_t12.log 2>&1; echo rc=$? >> gpurun_out/r2_t12.log; tail -4 gpurun_out/r2_t12.log
LCAO_K2_PIPE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r2_b12.json 2> gpurun_out/r2_b12.err; echo bench rc=$?
