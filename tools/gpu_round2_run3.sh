set -x
python -m pytest tests -m gpu -q -x > gpurun_out/r2_t3.log 2>&1; echo rc=$? >> gpurun_out/r2_t3.log; tail -3 gpurun_out/r2_t3.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r2_b3.json 2> gpurun_out/r2_b3.err; echo bench rc=$?
ps -u $(id -un) -o pid,ppid,etime,cmd > gpurun_out/r2_ps_after_bench.txt
python bench.py --workload cfg4 --emulate-world 8 --steps 5 --warmup 2 > gpurun_out/r2_cfg4_emu.json 2> gpurun_out/r2_cfg4_emu.err; echo emu rc=$?
python bench.py --scaling strong --steps 20 --warmup 5 > gpurun_out/r2_strong1.json 2> gpurun_out/r2_strong1.err; echo strong rc=$?
LCAO_BENCH_REHEARSAL=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --workload cfg4 --n-side 40 --steps 3 --warmup 1 > gpurun_out/r2_cfg4_reh2.json 2> gpurun_out/r2_cfg4_reh2.err; echo reh2 rc=$?
LCAO_BENCH_REHEARSAL=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 4 --workload cfg4 --n-side 64 --steps 3 --warmup 1 > gpurun_out/r2_cfg4_reh4.json 2> gpurun_out/r2_cfg4_reh4.err; echo reh4 rc=$?
LCAO_BENCH_REHEARSAL=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --scaling strong --steps 3 --warmup 1 > gpurun_out/r2_strong_reh2.json 2> gpurun_out/r2_strong_reh2.err; echo sreh rc=$?
