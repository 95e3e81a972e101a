set -x
timeout -k 10 600 python -m pytest tests/test_gpu_double.py tests/test_gpu_ddp.py -q -x -s > gpurun_out/r2_t9.log 2>&1; echo rc=$? >> gpurun_out/r2_t9.log; tail -12 gpurun_out/r2_t9.log
grep -q "rc=0" gpurun_out/r2_t9.log && timeout -k 10 120 python tools/probe_csr_capture.py C > gpurun_out/csr_probe/C.txt 2>&1; echo C rc=$? >> gpurun_out/csr_probe/C.txt; tail -6 gpurun_out/csr_probe/C.txt
