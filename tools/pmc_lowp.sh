# PMC passes over tools/k2_lowp_time.py for one variant library (gpurun box).  usage: bash tools/pmc_lowp.sh <lib glob> <tag>
LIB=$1; TAG=$2
OUT=$PWD/gpurun_out/pmc_lowp_$TAG
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/sq1 -o s -- python3 $REPO/tools/k2_lowp_time.py "$REPO/$LIB" > $OUT/sq1.log 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD --kernel-trace -d $OUT/sq2 -o s -- python3 $REPO/tools/k2_lowp_time.py "$REPO/$LIB" > $OUT/sq2.log 2>&1
cd $REPO
python3 tools/pmc_summary.py $OUT/sq1 coeff_mlp > $OUT/sq1_summary.txt 2>&1
python3 tools/pmc_summary.py $OUT/sq2 coeff_mlp > $OUT/sq2_summary.txt 2>&1
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*_counter_collection.csv" -delete; find $OUT -name "*.db" -delete
cat $OUT/sq1_summary.txt $OUT/sq2_summary.txt
