#!/bin/bash
# Runs ON THE GPU BOX: everything profiles/ keeps for a round - rocprofv3 kernel stats + PMC passes for config 2 (opacity A and B),
# config 4, the 1M dense pass and the one-view shape; the lane counts of the counting build; the side benches.
#   usage: tools/profile_round.sh <round tag, e.g. r03>        (outputs under gpurun_out/<tag>_*)
R=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
tools/prof.sh ${R}_c2 > /dev/null 2>&1
tools/prof.sh ${R}_c2b --opacity B > /dev/null 2>&1
tools/prof.sh ${R}_c4 --config C4 > /dev/null 2>&1
PROF_CMD="python $ROOT/tools/big_case.py" tools/prof.sh ${R}_dense > /dev/null 2>&1
PROF_CMD="python $ROOT/tools/prof_single_view.py" tools/prof.sh ${R}_single_view > /dev/null 2>&1
PROF_CMD="python $ROOT/tools/prof_dense_1v.py 12" tools/prof.sh ${R}_dense_1v > /dev/null 2>&1      # bench.py's dense_1m: ONE view per call (long-tile kernels)
PROF_CMD="python $ROOT/tools/prof_dense_1v.py 4" tools/prof.sh ${R}_dense_1v_cam4 > /dev/null 2>&1
PROF_CMD="python $ROOT/tools/loss_prof.py" tools/prof.sh ${R}_loss > /dev/null 2>&1                  # fused photometric loss, 24 x 512^2
( export TMPDIR=/tmp; cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/${R}_gi -o t -- python $ROOT/tools/prof_graphed_iteration.py > /dev/null 2>&1; cp /tmp/${R}_gi/t_kernel_stats.csv $ROOT/gpurun_out/${R}_graphed_iteration_kernel_stats.csv )   # the launches of one replayed iteration
python tools/sweep_loss_kernels.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_loss_sweep.txt
python tools/experiments/counting_build.py > /dev/null 2>&1       # the shipped sources carry no counting hooks: an instrumented copy
for a in "C2 A" "C2 B" "C4 A"; do T4D_LIB=$ROOT/topo4d_amd/csrc/variants/lib_count.so python tools/count_lanes.py $a --merge 2>/dev/null | tail -1; done > gpurun_out/${R}_lanes.jsonl
# the counters go into profiles/*.json HERE too (the box's copy), so that the bench lines below carry them as "current";
# profiles/ does not travel back: the same merges are repeated in the repository from gpurun_out/ (see profiles/README.md)
python tools/merge_counters.py gpurun_out/${R}_c2 C2 > /dev/null
python tools/merge_counters.py gpurun_out/${R}_c2b C2_B > /dev/null
python tools/merge_counters.py gpurun_out/${R}_c4 C4 > /dev/null
python tools/merge_counters.py gpurun_out/${R}_dense DENSE_1M > /dev/null
python tools/merge_counters.py gpurun_out/${R}_single_view SINGLE_VIEW > /dev/null
python tools/merge_counters.py gpurun_out/${R}_dense_1v DENSE_1M_1V > /dev/null
cp profiles/traffic.json profiles/valu.json profiles/lanes.json gpurun_out/ 2>/dev/null
tools/side_benches.sh ${R}_side > /dev/null 2>&1
python tools/small_launch.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_small_launch.txt      # 1-8 views per call, every build switch
tools/micro/sort_bin > gpurun_out/${R}_sort_bin_micro.txt 2>&1
python bench.py > gpurun_out/${R}_bench_c2.json 2> gpurun_out/${R}_bench_c2.err
python bench.py --config C4 --steps 20 > gpurun_out/${R}_bench_c4.json 2> gpurun_out/${R}_bench_c4.err
ls gpurun_out | grep ${R}_
