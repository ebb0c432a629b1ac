B="--steps 10 --warmup 3 --no-collate --no-cpu-baseline --no-vocoder --no-kernel-profile"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["value"])'
for n in 256 512 1024; do echo "=== LN blocks $n"; A3T_LN_BLOCKS=$n python bench.py $B 2>/dev/null | python -c "$P"; done
