# Diagnostic: rocprofv3 --pmc <counters...> over a bench.py run of the current build; prints per-launch means of the render kernel.
#   bash tools/pmc_probe.sh "" SQ_INSTS_VALU SQ_WAVES                                   (the headline workload)
#   bash tools/pmc_probe.sh "--scene dragons.json --width 3840 --height 2160" SQ_INSTS_VMEM
# (counters of one pass only: the SQ has 8 slots, FETCH_SIZE and WRITE_SIZE cannot share a pass - MI355X guide.)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ARGS=$1; shift
mkdir -p gpurun_out/probe
rm -rf gpurun_out/probe/p
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/probe/p -- python3 bench.py $ARGS --steps 10 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/probe/p.log 2>&1
python3 tools/pmc_summary.py gpurun_out/probe/p
