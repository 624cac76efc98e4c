set -e
cd $GRAFT_REPO_ROOT
python3 tools/variants.py "x0=" "x4=-DRTC_EXPERIMENT=4" -- python3 tools/time_scenes.py --scenes dragons,teapot,nefertiti > gpurun_out/r5_bound_fp32_tri.txt 2>&1
