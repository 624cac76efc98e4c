# Diagnostic: like pmc_probe.sh for another scene: pmc_probe_scene.sh <scene> <w> <h> <counters...>
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
SC=$1; W=$2; H=$3; shift 3
mkdir -p gpurun_out/probe
rm -rf gpurun_out/probe/p
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/probe/p -- python3 bench.py --scene $SC --width $W --height $H --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/probe/p.log 2>&1
python3 tools/pmc_summary.py gpurun_out/probe/p
