#!/bin/bash
# bench.py A/B of one environment variable on ONE box:  tools/ab_env.sh <workload> <steps> <VAR> <value> <value> ...
cd ${GRAFT_REPO_ROOT:-.}
w=$1; k=$2; var=$3; shift 3
for v in "$@"; do
  echo "== $w $var=$v"
  env $var=$v python bench.py --workload $w --steps $k --warmup 2 --no-cpu-baseline --no-extra --no-roofline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'], r['timing']['ms_per_step_min'])"
done
