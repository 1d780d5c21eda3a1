# usage (on the GPU box): bash tools/step_ab.sh ENVVAR VALUE_A VALUE_B [pairs]  -- same-box A/B of `python bench.py --steps 10 --warmup 3`
# under an environment variable (library options read theirs once per process), alternating, `pairs` times (default 3)
R=$GRAFT_REPO_ROOT
V=$1; A=$2; B=$3; N=${4:-3}
for i in $(seq 1 $N); do
  for val in $A $B; do
    env $V=$val python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline())
print('$V=$val', d['ms_per_step'], 'ms/step', d['value'], 'img/s', 'clock', d['clock_state']['probe_ghz'], 'conv8', (d.get('roofline_conv8') or {}).get('frac'), 'up', (d.get('roofline_up') or {}).get('achieved_per_launch_ms'))"
  done
done
