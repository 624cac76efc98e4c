# One-GPU rehearsal of the N-way split with 1, 2, 3 ... frames in flight per rank (tools/scale_sim.py --inflight): bash tools/scale_sim_inflight.sh "1 2 3" > out.jsonl
cd $GRAFT_REPO_ROOT
for m in ${1:-1 2 3}; do
  for a in "" "--scene teapot.json" "--scene dragons.json --width 3840 --height 2160"; do
    python3 tools/scale_sim.py $a --tiles 64 --worlds ${2:-4,8} --reps 20 --inflight $m
  done
done
