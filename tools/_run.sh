set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 -L > gpurun_out/r5_counters_list.txt 2>&1 || true
grep -i -o "SQC_[A-Z_0-9]*\|SQ_IFETCH[A-Z_0-9]*\|SQ_INST_CYCLES[A-Z_0-9]*\|SQ_WAIT_IFETCH[A-Z_]*" gpurun_out/r5_counters_list.txt | sort -u > gpurun_out/r5_sqc_names.txt
bash tools/pmc_probe.sh "--scene dragons.json --width 3840 --height 2160 --option waves3=1" SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY > gpurun_out/r5_icache_dragons.txt 2>&1
bash tools/pmc_probe.sh "" SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY > gpurun_out/r5_icache_cover.txt 2>&1
