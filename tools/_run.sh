set -e
cd $GRAFT_REPO_ROOT
python3 tools/variants.py "base=" "O2=-O2" "maxilp=-mllvm -amdgpu-sched-strategy=max-ilp" "maxmem=-mllvm -amdgpu-sched-strategy=max-memory-clause" "nopost=-mllvm -enable-post-misched=0" "sink=-mllvm -sink-insts-to-avoid-spills" -- python3 tools/time_scenes.py --scenes cover,dragons,teapot,nefertiti --check > gpurun_out/r5_flags.txt 2>&1 || true
