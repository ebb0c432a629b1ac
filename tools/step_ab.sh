#!/bin/bash
# step time under environment-variable variants, same box: tools/step_ab.sh "NAME:VAR=val VAR=val" ...
cd "$(dirname "$0")/.."
for v in "$@"; do
  name=${v%%:*}; envs=${v#*:}
  out=$(env $envs python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-vocoder --no-collate --no-c4 --no-kernel-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%.2f ms/step' % d['ms_per_step'])")
  echo "$name [$envs]: $out"
done
