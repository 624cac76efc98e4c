set -e
cd $GRAFT_REPO_ROOT
python3 tools/variants.py "dp20=-DRTC_COLLAPSE_DP=1 -DRTC_COLLAPSE_PRIM_COST=2.0" "dp40=-DRTC_COLLAPSE_DP=1 -DRTC_COLLAPSE_PRIM_COST=4.0" "dp100=-DRTC_COLLAPSE_DP=1 -DRTC_COLLAPSE_PRIM_COST=10.0" "dp1000=-DRTC_COLLAPSE_DP=1 -DRTC_COLLAPSE_PRIM_COST=1000.0" -- python3 tools/time_scenes.py --scenes dragons,teapot,nefertiti,groups --check > gpurun_out/r5_dp2.txt 2>&1
