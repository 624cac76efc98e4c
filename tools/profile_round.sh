# Collects the rocprofv3 evidence for one round on the GPU box: kernel-trace stats and, in separate
# passes (FETCH_SIZE and WRITE_SIZE cannot share a pass), the PMC counters; the same for dragons.json at 4K (the
# mesh path); all five BASELINE configs; the full bench line.  Usage: bash tools/profile_round.sh r02
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=${1:-r02}
OUT=gpurun_out/profiles_$R
mkdir -p $OUT
CMD="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq1 -- $CMD > $OUT/pmc_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $OUT/pmc_sq2 -- $CMD > $OUT/pmc_sq2.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_EA0_WRREQ --output-format csv -d $OUT/pmc_tcc -- $CMD > $OUT/pmc_tcc.log 2>&1 || true
echo "cover passes done"
DCMD="python3 bench.py --scene dragons.json --width 3840 --height 2160 --steps 10 --warmup 2 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/d_stats -- $DCMD > $OUT/d_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/d_pmc_fetch -- $DCMD > $OUT/d_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/d_pmc_write -- $DCMD > $OUT/d_pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/d_pmc_sq1 -- $DCMD > $OUT/d_pmc_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_EA0_WRREQ --output-format csv -d $OUT/d_pmc_tcc -- $DCMD > $OUT/d_pmc_tcc.log 2>&1 || true
echo "dragons passes done"
python3 tools/bench_configs.py > $OUT/configs.txt 2> $OUT/configs.err
echo "configs done"
python3 bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
