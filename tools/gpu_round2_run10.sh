set -x
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r2_t10.log 2>&1; echo rc=$? >> gpurun_out/r2_t10.log; tail -4 gpurun_out/r2_t10.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r2_b10.json 2> gpurun_out/r2_b10.err; echo bench rc=$?
python bench.py --workload cfg4 --emulate-world 8 --steps 5 --warmup 2 > gpurun_out/r2_cfg4_emu10.json 2> gpurun_out/r2_cfg4_emu10.err; echo emu rc=$?
LCAO_BENCH_REHEARSAL=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 4 --workload cfg4 --n-side 64 --steps 3 --warmup 1 > gpurun_out/r2_cfg4_reh4b.json 2> gpurun_out/r2_cfg4_reh4b.err; echo reh4 rc=$?
python bench.py --workload cfg5-1of8 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2_cfg5.json 2> gpurun_out/r2_cfg5.err; echo cfg5 rc=$?
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2_smoke.txt 2>&1; echo smoke rc=$?
