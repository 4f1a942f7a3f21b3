#!/bin/bash
# In the repository, after `gpurun -- bash tools/profile_round.sh <R>`: copies what profiles/ keeps of gpurun_out/<R>_* and repeats
# the counter merges (profiles/{traffic,valu,lanes}.json carry the kernel-source hash bench.py checks).   usage: tools/collect_profiles.sh r05
R=${1:-rXX}
cd "$(dirname "$0")/.."
G=gpurun_out; P=profiles
cp $G/${R}_c2/kernel_stats.txt $P/${R}_kernel_stats.txt; cp $G/${R}_c2/pmc_summary.txt $P/${R}_pmc_summary.txt; cp $G/${R}_c2/step_timeline.txt $P/${R}_step_timeline.txt
for c in c2b c4; do for f in kernel_stats pmc_summary step_timeline; do cp $G/${R}_$c/$f.txt $P/${R}_${c}_$f.txt; done; done
cp $G/${R}_dense/kernel_stats.txt $P/${R}_dense_1M_kernel_stats.txt; cp $G/${R}_dense/pmc_summary.txt $P/${R}_dense_1M_pmc_summary.txt
cp $G/${R}_single_view/kernel_stats.txt $P/${R}_single_view_kernel_stats.txt; cp $G/${R}_single_view/pmc_summary.txt $P/${R}_single_view_pmc_summary.txt
for c in dense_1v dense_1v_cam4; do for f in kernel_stats pmc_summary step_timeline; do cp $G/${R}_$c/$f.txt $P/${R}_${c}_$f.txt; done; done
cp $G/${R}_loss/kernel_stats.txt $P/${R}_loss_kernel_stats.txt; cp $G/${R}_loss/pmc_summary.txt $P/${R}_loss_pmc_summary.txt
cp $G/${R}_side/big_case.txt $P/${R}_big_case_1M_4096x3008.txt; cp $G/${R}_side/single_view.txt $P/${R}_single_view_kernels.txt
cp $G/${R}_side/bench_dropin.json $P/${R}_bench_dropin.json; cp $G/${R}_side/bench_bake.json $P/${R}_bench_bake_8192.json; cp $G/${R}_side/bench_loss.json $P/${R}_bench_loss.json
for f in loss_sweep.txt small_launch.txt sort_bin_micro.txt lanes.jsonl bench_c2.json bench_c4.json; do cp $G/${R}_$f $P/${R}_$f; done
python tools/merge_counters.py $G/${R}_c2 C2 > /dev/null; python tools/merge_counters.py $G/${R}_c2b C2_B > /dev/null; python tools/merge_counters.py $G/${R}_c4 C4 > /dev/null
python tools/merge_counters.py $G/${R}_dense DENSE_1M > /dev/null; python tools/merge_counters.py $G/${R}_single_view SINGLE_VIEW > /dev/null
python tools/merge_counters.py $G/${R}_dense_1v DENSE_1M_1V > /dev/null
python - <<PY
import json
cur = json.load(open("$P/lanes.json"))
for l in open("$G/${R}_lanes.jsonl"):
    d = json.loads(l); cur[d["config"] + ("" if d["opacity"] == "A" else "_" + d["opacity"])] = d
json.dump(cur, open("$P/lanes.json", "w"), indent=1)
PY
git status --short $P | head -60
