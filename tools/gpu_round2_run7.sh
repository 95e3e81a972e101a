set -x
python -m pytest tests -m gpu -q > gpurun_out/r2_t7.log 2>&1; echo rc=$? >> gpurun_out/r2_t7.log; tail -4 gpurun_out/r2_t7.log
timeout -k 10 120 python tools/probe_csr_capture.py B > gpurun_out/csr_probe/B.txt 2>&1; echo B rc=$? >> gpurun_out/csr_probe/B.txt; tail -5 gpurun_out/csr_probe/B.txt
