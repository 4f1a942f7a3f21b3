#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the side measurements DESIGN.md quotes, one output file each under gpurun_out/<tag>/
# (copy the ones worth keeping into profiles/).   usage: tools/side_benches.sh <tag>
TAG=${1:-side}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
python tools/bench_bake.py            > $OUT/bench_bake.json   2> $OUT/bench_bake.err
python tools/bench_loss.py            > $OUT/bench_loss.json   2> $OUT/bench_loss.err
python tools/bench_dropin.py          > $OUT/bench_dropin.json 2> $OUT/bench_dropin.err
python tools/prof_single_view.py      > $OUT/single_view.txt   2> $OUT/single_view.err
python tools/big_case.py              > $OUT/big_case.txt      2> $OUT/big_case.err
tail -n 3 $OUT/*.json $OUT/*.txt
