# rocprofv3 evidence for the current kernel state (run on the GPU box through gpurun; outputs under gpurun_out/prof_<tag>)
# usage: bash tools/profile_round.sh <tag> <commit> <date>
set -x
TAG=$1; COMMIT=$2; DATE=$3
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/kt -o kt -- python3 $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/kt_bench.json 2> $OUT/kt_bench.err
rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o f -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/f_bench.json 2> $OUT/f_bench.err
rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o w -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/w_bench.json 2> $OUT/w_bench.err
rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace -d $OUT/sq -o s -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/s_bench.json 2> $OUT/s_bench.err
cd $REPO
python3 tools/pmc_summary.py $OUT/fetch > $OUT/pmc_fetch_summary.txt 2>&1
python3 tools/pmc_summary.py $OUT/write > $OUT/pmc_write_summary.txt 2>&1
python3 tools/pmc_summary.py $OUT/sq > $OUT/pmc_sq_summary.txt 2>&1
python3 tools/pmc_traffic.py $OUT/fetch $OUT/write $OUT/pmc_traffic.json $COMMIT $DATE > /dev/null 2> $OUT/pmc_traffic.err
find $OUT/kt -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
# keep the merge-back small: the raw traces are large
find $OUT -name "*_kernel_trace.csv" -size +4M -delete
find $OUT -name "*_counter_collection.csv" -size +8M -delete
find $OUT -name "*.db" -delete
ls -la $OUT
