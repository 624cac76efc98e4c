cd $GRAFT_REPO_ROOT
for o in ${OPTS:-cut_above=0}; do
  for a in "" "--scene dragons.json --width 3840 --height 2160" "--scene teapot.json"; do
    python3 tools/scale_sim.py $a --tiles 64 --worlds ${WORLDS:-8} --reps 20 --inflight 3 --option $o 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l)
    if 'scene' in d: print('$o', d['scene'], 'full %.3f' % d['full_ms'], end=' ')
    else: print(d['world'], 'way max %.3f' % d['by_measured_cost']['max_ms'], end=' ')
print()
"
  done
done
