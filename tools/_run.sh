set -e
cd $GRAFT_REPO_ROOT
python3 tools/time_scenes.py --set all --check > gpurun_out/r5_box4.txt 2>&1
python -m pytest tests -m gpu -x -q > gpurun_out/r5_gpu5.log 2>&1 || true
tail -5 gpurun_out/r5_gpu5.log
