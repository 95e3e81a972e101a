# PMC passes (SQ + LDS counters) over `bench.py --precision <p>` for the dominant kernel of a secondary mode.
# usage: bash tools/pmc_bench_mode.sh <precision> <tag>
PREC=$1; TAG=$2
OUT=$PWD/gpurun_out/pmc_mode_$TAG
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/kt -o kt -- python3 $REPO/bench.py --precision $PREC --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/kt_bench.json 2> $OUT/kt_bench.err
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/sq1 -o s -- python3 $REPO/bench.py --precision $PREC --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/sq1.json 2> $OUT/sq1.err
rocprofv3 --output-format csv --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD --kernel-trace -d $OUT/sq2 -o s -- python3 $REPO/bench.py --precision $PREC --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/sq2.json 2> $OUT/sq2.err
cd $REPO
python3 tools/pmc_summary.py $OUT/sq1 > $OUT/sq1_summary.txt 2>&1
python3 tools/pmc_summary.py $OUT/sq2 > $OUT/sq2_summary.txt 2>&1
find $OUT/kt -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*_counter_collection.csv" -delete; find $OUT -name "*.db" -delete
head -8 $OUT/kernel_stats.csv | cut -c1-150; head -12 $OUT/sq1_summary.txt
