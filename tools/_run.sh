set -e
cd $GRAFT_REPO_ROOT
( python3 tools/fuzz_more.py 40000 1200 5 2>&1 | tail -4 ) > gpurun_out/r5_fuzz_groups.txt
echo "groups depth 5 done"
( python3 tools/fuzz_more.py 50000 300 8 2>&1 | tail -3 ) >> gpurun_out/r5_fuzz_groups.txt
echo "groups depth 8 done"
( python3 tools/fuzz_flat.py 3000 400 2>&1 | tail -4 ) > gpurun_out/r5_fuzz_flat.txt || true
echo "flat done"
( python3 tools/full_size_parity.py 2>&1 | tail -22 ) > gpurun_out/r5_full_size_parity.txt
echo "full size done"
