set -e
cd $GRAFT_REPO_ROOT
python3 tools/create_time.py > gpurun_out/r5_create_time.txt 2>&1
python3 tools/variants.py "lite=-DRTC_PROFILE -DRTC_PROFILE_LITE" -- python3 tools/wave_ends.py cover 1920 1080 5 -- python3 tools/wave_ends.py dragons 3840 2160 5 -- python3 tools/wave_ends.py teapot 1920 1080 5 > gpurun_out/r5_wave_ends.txt 2>&1
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r5_bench_quick.json 2> gpurun_out/r5_bench_quick.err
