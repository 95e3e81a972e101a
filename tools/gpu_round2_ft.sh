set -x
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_double.py -m gpu -q -x > gpurun_out/r2_ft_t.log 2>&1 && \
tail -3 gpurun_out/r2_ft_t.log && \
timeout -k 10 400 python tools/time_force_training.py --n-cells 250 --torch-profile --out gpurun_out/ft_250f.json > gpurun_out/ft_250f.log 2>&1; echo rc=$?; tail -3 gpurun_out/r2_ft_t.log
