# The DYNAMIC instruction mix of a bench command's render kernel from the SQ's own per-class counters (no instrumentation):
#   bash tools/pmc_mix.sh <out-name> [bench.py arguments]      ->  gpurun_out/<out-name>.txt
# Three rocprofv3 --pmc passes (FP64 classes; FP32 classes; integer / conversion / scalar / memory), per-launch means.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
NAME=$1; shift
OUT=gpurun_out/pmc_mix_$NAME
mkdir -p $OUT
CMD="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras $@"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_BUSY_CYCLES --output-format csv -d $OUT/a -- $CMD > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_BRANCH --output-format csv -d $OUT/b -- $CMD > $OUT/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/c -- $CMD > $OUT/c.log 2>&1
python3 tools/pmc_mix.py $OUT > gpurun_out/$NAME.txt
cat gpurun_out/$NAME.txt
