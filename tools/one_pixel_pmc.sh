# PMC passes over a launch that is ONE wave's work (a single pixel or small rectangle): instructions, waits, I-cache.
#   bash tools/one_pixel_pmc.sh <name> cover.json 1920 1080 X Y W H [depth]
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
NAME=$1; shift
OUT=gpurun_out/pmc_one_$NAME
mkdir -p $OUT
CMD="python3 tools/one_pixel_render.py $@"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/a -- $CMD > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/b -- $CMD > $OUT/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_BRANCH SQ_WAVES --output-format csv -d $OUT/c -- $CMD > $OUT/c.log 2>&1
python3 - <<PY
import sys, os
sys.path.insert(0, "tools")
from pmc_summary import summarize
c = summarize(["$OUT/a", "$OUT/b", "$OUT/c"], kernel="rtc_render_kernel")
for k in sorted(c): print(f"{k:28s} {c[k]:14.1f}")
PY
