set -e
cd $GRAFT_REPO_ROOT
python3 tools/time_scenes.py --scenes cover,cubes,dragons,groups,cylinders > gpurun_out/r5_box5.txt 2>&1
bash tools/pmc_probe.sh "" WRITE_SIZE >> gpurun_out/r5_box5.txt 2>&1
bash tools/pmc_probe.sh "--option box_cull=0" WRITE_SIZE >> gpurun_out/r5_box5.txt 2>&1
