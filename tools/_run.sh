set -e
cd $GRAFT_REPO_ROOT
python3 tools/variants.py "prof=-DRTC_PROFILE" -- python3 tools/prof_sections.py dragons.json 3840 2160 > gpurun_out/r5_dark_prof.txt 2>&1
