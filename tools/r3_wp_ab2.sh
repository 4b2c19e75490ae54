#!/bin/bash
# as r3_wp_ab.sh plus C4 B=256 and family evaluation
for v in "$@"; do
  L=$PWD/red-gnn_amd/libredgnn_$v.so
  for cfg in "C3 256" "C2 1024" "C4 256" "C2 64"; do set -- $cfg
    RG_LIB=$L python bench.py --config $1 --batch $2 --steps 8 --warmup 2 --no-cpu-baseline --no-family-eval --no-dense-f32 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v $1 B=$2 step %.3f ms'%d['ms_per_step'], ['%.3f'%h['ms'] for h in d['per_hop']])"
  done
  RG_LIB=$L python tools/probe_eval_family.py 2>/dev/null | head -1
done
