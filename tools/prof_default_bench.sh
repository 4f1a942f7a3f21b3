# Runs ON THE GPU BOX: rocprofv3 --kernel-trace --stats of the DEFAULT bench command (three frames in flight at config 2): the kernels of the
# streams overlap, so a duration here includes the time a kernel shares the chip -> gpurun_out/r03_kernel_stats_three_in_flight.txt
export TMPDIR=/tmp; ROOT=${GRAFT_REPO_ROOT:-/root/repo}; RAW=/tmp/prof_raw_3f; rm -rf $RAW; mkdir -p $RAW /tmp/prof_out_3f
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o t -- python $ROOT/bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 3 > /dev/null 2>&1
python $ROOT/tools/summarize_prof.py $RAW /tmp/prof_out_3f > /dev/null 2>&1
cp /tmp/prof_out_3f/kernel_stats.txt $ROOT/gpurun_out/r03_kernel_stats_three_in_flight.txt
head -10 /tmp/prof_out_3f/kernel_stats.txt
