# Diagnostic: rocprofv3 --pmc <counters...> over bench.py for the current build; prints per-launch means.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/probe
rm -rf gpurun_out/probe/p
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/probe/p -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/probe/p.log 2>&1
python3 tools/pmc_summary.py gpurun_out/probe/p
