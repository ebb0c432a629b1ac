#!/bin/bash
# step time under environment-variable variants, same box: tools/step_ab.sh "NAME:VAR=val VAR=val" ...
cd "$(dirname "$0")/.."
for v in "$@"; do
  name=${v%%:*}; envs=${v#*:}
  out=$(env $envs python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-vocoder --no-collate 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('%.2f ms/step  all_gemms_alone %.2f ms  dominant %s %.0f us' % (d['ms_per_step'], r.get('all_gemms_alone',{}).get('ms',0), r['kernel'], r['avg_us']))")
  echo "$name [$envs]: $out"
done
