# Collects the rocprofv3 evidence for one round on the GPU box.  Per workload (cover 1080p - the headline -, dragons 4K and
# teapot 1080p - the mesh path): kernel-trace stats and, in separate passes (FETCH_SIZE and WRITE_SIZE cannot share a
# pass), the PMC counters incl. the per-class instruction counters; then all five BASELINE configs, the one-GPU rehearsal
# of the N-way split, and the full bench lines.  Usage: bash tools/profile_round.sh r03
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=${1:-r05}
OUT=gpurun_out/profiles_$R
mkdir -p $OUT
passes() {  # passes <prefix> <bench arguments...>
  P=$1; shift
  CMD="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras $@"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${P}stats -- $CMD > $OUT/${P}stats.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${P}pmc_fetch -- $CMD > $OUT/${P}pmc_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${P}pmc_write -- $CMD > $OUT/${P}pmc_write.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/${P}pmc_sq1 -- $CMD > $OUT/${P}pmc_sq1.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $OUT/${P}pmc_sq2 -- $CMD > $OUT/${P}pmc_sq2.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_BRANCH --output-format csv -d $OUT/${P}pmc_sq3 -- $CMD > $OUT/${P}pmc_sq3.log 2>&1
  rocprofv3 --kernel-trace --pmc TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_EA0_WRREQ --output-format csv -d $OUT/${P}pmc_tcc -- $CMD > $OUT/${P}pmc_tcc.log 2>&1 || true
}
passes ""
echo "cover passes done"
# (dragons: its handles choose the three-wave kernel - profiles/r04/bvh8_experiments.md; pinned here so that every pass counts the same kernel)
passes d_ --scene dragons.json --width 3840 --height 2160 --option waves3=1
echo "dragons passes done"
passes t_ --scene teapot.json --option waves3=0
echo "teapot passes done"
python3 tools/bench_configs.py > $OUT/configs.txt 2> $OUT/configs.err
echo "configs done"
for s in cover dragons teapot; do
  case $s in dragons) A="--scene dragons.json --width 3840 --height 2160";; teapot) A="--scene teapot.json";; *) A="";; esac
  python3 tools/scale_sim.py $A --tiles 64 --reps 20 >> $OUT/scale_sim.txt 2>> $OUT/scale_sim.err || true
  python3 tools/scale_sim.py $A --tiles 64 --worlds 4,8 --reps 20 --inflight 3 >> $OUT/scale_sim_inflight3.txt 2>> $OUT/scale_sim.err || true
done
echo "scale_sim done"
