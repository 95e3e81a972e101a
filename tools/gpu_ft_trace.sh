# kernel trace of the force-training step (tools/time_force_training.py) and its idle gaps; usage: bash tools/gpu_ft_trace.sh <tag>
TAG=$1
OUT=$PWD/gpurun_out/ft_trace_$TAG
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/kt -o kt -- python3 $REPO/tools/time_force_training.py --steps 4 --out $OUT/ft.json > $OUT/ft.log 2>&1
cd $REPO
python3 tools/trace_gaps.py $OUT/kt 0.6 > $OUT/gaps.txt 2>&1
find $OUT/kt -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
head -24 $OUT/gaps.txt
