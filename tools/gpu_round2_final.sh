set -x
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r2_tfinal.log 2>&1; echo rc=$? >> gpurun_out/r2_tfinal.log; tail -4 gpurun_out/r2_tfinal.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2_smoke_final.txt 2>&1; echo smoke rc=$?
python bench.py > gpurun_out/r2_bfinal.json 2> gpurun_out/r2_bfinal.err; echo bench rc=$?
