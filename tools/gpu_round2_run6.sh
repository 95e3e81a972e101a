set -x
python -m pytest tests -m gpu -q > gpurun_out/r2_t6.log 2>&1; echo rc=$? >> gpurun_out/r2_t6.log; tail -4 gpurun_out/r2_t6.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r2_b6.json 2> gpurun_out/r2_b6.err; echo bench rc=$?
mkdir -p gpurun_out/csr_probe
timeout -k 10 120 python tools/probe_csr_capture.py A > gpurun_out/csr_probe/A.txt 2>&1; echo A rc=$? >> gpurun_out/csr_probe/A.txt; tail -5 gpurun_out/csr_probe/A.txt
