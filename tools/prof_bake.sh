#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel-trace stats of the 8192^2 bake (tools/bench_bake.py --no-cpu), summary to
# gpurun_out/<tag>_bake_kernel_stats.txt.     usage: tools/prof_bake.sh <tag>
TAG=$1; shift
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
RAW=/tmp/prof_raw_bake_$TAG
rm -rf $RAW; mkdir -p $RAW /tmp/prof_out_bake_$TAG $ROOT/gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o t -- python $ROOT/tools/bench_bake.py --no-cpu "$@" > /dev/null 2>&1
python $ROOT/tools/summarize_prof.py $RAW /tmp/prof_out_bake_$TAG > /dev/null 2>&1
cp /tmp/prof_out_bake_$TAG/kernel_stats.txt $ROOT/gpurun_out/${TAG}_bake_kernel_stats.txt
head -14 /tmp/prof_out_bake_$TAG/kernel_stats.txt
