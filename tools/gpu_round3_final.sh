# Round-end evidence, part A: the full GPU test suite, smoke, the full bench line and the force-training breakdowns.
set -x
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r3_tfinal.log 2>&1; echo rc=$? >> gpurun_out/r3_tfinal.log; tail -3 gpurun_out/r3_tfinal.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3_smoke_final.txt 2>&1; echo smoke rc=$?
python bench.py > gpurun_out/r3_bfinal.json 2> gpurun_out/r3_bfinal.err; echo bench rc=$?
python tools/time_force_training.py --torch-profile --out gpurun_out/r3_force_training_1000.json > gpurun_out/r3_ft.log 2>&1
python tools/time_force_training.py --chunks 2 --out gpurun_out/r3_force_training_1000_2groups.json > gpurun_out/r3_ft2.log 2>&1
python tools/time_trip_family.py > gpurun_out/r3_force_training_kernels.txt 2>&1
