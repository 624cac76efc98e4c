set -e
cd $GRAFT_REPO_ROOT
python3 tools/variants.py "base=-DRTC_ROOT_NODE_IN_REC=0" "rootrec=" -- python3 tools/time_scenes.py --scenes dragons,teapot,nefertiti,groups,csg_demo --check -- python3 tools/time_scenes.py --scenes dragons,teapot,nefertiti --option waves3=1 -- python3 tools/time_scenes.py --scenes dragons,teapot,nefertiti --option waves3=0 > gpurun_out/r5_rootrec.txt 2>&1
