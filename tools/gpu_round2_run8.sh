set -x
mkdir -p gpurun_out/csr_probe
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r2_t8.log 2>&1; echo rc=$? >> gpurun_out/r2_t8.log; tail -4 gpurun_out/r2_t8.log
grep -q "rc=0" gpurun_out/r2_t8.log && python bench.py --steps 20 --warmup 5 > gpurun_out/r2_b8.json 2> gpurun_out/r2_b8.err; echo bench rc=$?
grep -q "rc=0" gpurun_out/r2_t8.log && timeout -k 10 120 python tools/probe_csr_capture.py B > gpurun_out/csr_probe/B.txt 2>&1; echo B rc=$? >> gpurun_out/csr_probe/B.txt; tail -5 gpurun_out/csr_probe/B.txt
