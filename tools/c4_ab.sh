#!/bin/bash
# configs[3] (d=512, H=4, x-vector, B=16, T_mel=1600) step time under environment variants: tools/c4_ab.sh "NAME:VAR=val" ...
cd "$(dirname "$0")/.."
for v in "$@"; do
  name=${v%%:*}; envs=${v#*:}
  out=$(env $envs python -c "
import torch, bench, json
r = bench.c4_leg(torch.device('cuda', 0), 'bf16', steps=8, warmup=3)
print('%.2f ms/step' % r['ms_per_step'])" 2>/dev/null | tail -1)
  echo "$name [$envs]: $out"
done
