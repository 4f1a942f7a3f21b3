#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel trace + PMC passes over bench.py (or over PROF_CMD), summarised to small
# text files under gpurun_out/<tag>/ (copy the ones worth keeping into profiles/).
# usage: tools/prof.sh <tag> [bench args]          PROF_CMD="python tools/big_case.py" tools/prof.sh <tag>
TAG=$1; shift
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
RAW=/tmp/prof_raw_$TAG
rm -rf $RAW; mkdir -p $OUT $RAW
cd /tmp
if [ -n "$PROF_CMD" ]; then
  B="$PROF_CMD"; LONG=""; SHORT=""
else
  B="python $ROOT/bench.py --no-cpu-baseline --no-extras --frames-in-flight 1"     # kernels alone on the chip: per-kernel figures
  LONG="--steps 20 --warmup 3"; SHORT="--steps 2 --warmup 1"
fi
( cd $ROOT && $B $LONG "$@" ) > /dev/null 2>&1      # (first run outside the profiler: pages the image in)
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o t -- $B $LONG "$@" > $OUT/bench_trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $RAW/pmc1 -o p -- $B $SHORT "$@" > $OUT/bench_pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $RAW/pmc2 -o p -- $B $SHORT "$@" > $OUT/bench_pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $RAW/pmc3 -o p -- $B $SHORT "$@" > $OUT/bench_pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $RAW/pmc4 -o p -- $B $SHORT "$@" > $OUT/bench_pmc4.log 2>&1
python $ROOT/tools/summarize_prof.py $RAW $OUT > /dev/null
python - <<PY
import json, sys
sys.path.insert(0, "$ROOT")
from topo4d_amd.build import raster_source_sha256
p = "$OUT/counters.json"
c = json.load(open(p))
c["_kernel_source_sha256"] = raster_source_sha256()
json.dump(c, open(p, "w"), indent=1)
PY
ls $OUT
