# The vector-memory side of a bench command's render kernel: texture-addresser / L1 (TA, TCP) busy and stall counters.
#   bash tools/pmc_mem.sh <out-name> [bench.py arguments]      ->  gpurun_out/<out-name>.txt
# Separate rocprofv3 --pmc passes (few counters each: the TA / TCP blocks have few slots), per-launch means.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
NAME=$1; shift
OUT=gpurun_out/pmc_mem_$NAME
mkdir -p $OUT
CMD="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras $@"
rocprofv3 --list-avail > $OUT/avail.txt 2>&1 || true
i=0
while read -r SET; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1 && echo "pass $i done" || {
    rc=$?; echo "pass $i ($SET) failed rc=$rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "a pass that had to be killed ends the run"; break; fi
  }
done <<'SETS'
SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY
TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max
TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum
TA_FLAT_WRITE_WAVEFRONTS_sum TA_BUFFER_WRITE_WAVEFRONTS_sum
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum
TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
TCP_GATE_EN1_sum TCP_GATE_EN2_sum
TCP_TA_TCP_STATE_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum
TD_TD_BUSY_sum TD_TC_STALL_sum
SETS
python3 tools/pmc_mem.py $OUT $i > gpurun_out/$NAME.txt
cat gpurun_out/$NAME.txt
