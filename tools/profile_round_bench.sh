# Second half of a round's evidence: the full bench lines (they read profiles/traffic.json, which
# tools/collect_profiles.py writes from the passes of tools/profile_round.sh - so: profile_round.sh, collect, THEN this).
# Usage: bash tools/profile_round_bench.sh r03
set -e
cd $GRAFT_REPO_ROOT
R=${1:-r05}
OUT=gpurun_out/profiles_$R
mkdir -p $OUT
python3 bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
python3 bench.py --scene dragons.json --width 3840 --height 2160 --steps 20 --warmup 3 > $OUT/bench_dragons.json 2> $OUT/bench_dragons.err
python3 bench.py --scene teapot.json --steps 20 --warmup 3 > $OUT/bench_teapot.json 2> $OUT/bench_teapot.err
python3 bench.py --scene reflection_and_refraction.json --depth 8 --steps 20 --warmup 3 --no-extras > $OUT/bench_rr.json 2> $OUT/bench_rr.err
cat $OUT/bench.json | cut -c1-400
