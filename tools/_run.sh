set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r5_gpu6.log 2>&1 || true
tail -4 gpurun_out/r5_gpu6.log
( python3 tools/fuzz_flat.py 9000 500 7 box_cull=1 2>&1 | tail -3 ) > gpurun_out/r5_fuzz_box_simple.txt
( python3 tools/fuzz_flat.py 9500 300 7 box_cull=1 simple3_min_chunks=0 2>&1 | tail -3 ) >> gpurun_out/r5_fuzz_box_simple.txt
cat gpurun_out/r5_fuzz_box_simple.txt
