set -e
cd $GRAFT_REPO_ROOT
python3 tools/variants.py "base=-DRTC_MAJOR_XF=0" "mxf=" "leaf128=-DRTC_MAJOR_XF=0 -DRTC_LEAF_PAD_WORDS=8" "leaf96=-DRTC_MAJOR_XF=0 -DRTC_LEAF_PAD_WORDS=0" "node96=-DRTC_MAJOR_XF=0 -DRTC_NODE_PAD_BYTES=16" "node128=-DRTC_MAJOR_XF=0 -DRTC_NODE_PAD_BYTES=48" -- python3 tools/time_scenes.py --scenes dragons,teapot,nefertiti,groups --check > gpurun_out/r5_layout.txt 2>&1
