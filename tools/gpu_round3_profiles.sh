# Round-end evidence, part B: rocprofv3 kernel stats + PMC passes of the headline bench, of the f16 mode, and the cfg5-1of8 line.
set -x
COMMIT=$1; DATE=$2
bash tools/profile_round.sh r3 $COMMIT $DATE > gpurun_out/prof_r3.log 2>&1
bash tools/pmc_bench_mode.sh f16 r3f16 > gpurun_out/pmc_f16_r3.log 2>&1
python bench.py --workload cfg5-1of8 --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/r3_cfg5.json 2> gpurun_out/r3_cfg5.err
ls gpurun_out/prof_r3 gpurun_out/pmc_mode_r3f16
