#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel-trace stats of one bench.py run (one frame in flight), summary to stdout and to
# gpurun_out/<tag>_kernel_stats.txt.     usage: tools/prof_kernels.sh <tag> [bench args]
TAG=$1; shift
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
RAW=/tmp/prof_raw_$TAG
rm -rf $RAW; mkdir -p $RAW /tmp/prof_out_$TAG $ROOT/gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o t -- python $ROOT/bench.py --no-cpu-baseline --no-extras --frames-in-flight 1 --steps 10 --warmup 2 "$@" > /dev/null 2>&1
python $ROOT/tools/summarize_prof.py $RAW /tmp/prof_out_$TAG > /dev/null 2>&1
cp /tmp/prof_out_$TAG/kernel_stats.txt $ROOT/gpurun_out/${TAG}_kernel_stats.txt
head -14 /tmp/prof_out_$TAG/kernel_stats.txt
