#!/bin/bash
# A/B of word-parallel kernel builds: per-hop layer times at C3 B=256 and C2 B=1024, family evaluation throughput
for v in "$@"; do
  L=$PWD/red-gnn_amd/libredgnn_$v.so
  for cfg in "C3 256" "C2 1024"; do set -- $cfg
    RG_LIB=$L python bench.py --config $1 --batch $2 --steps 8 --warmup 2 --no-cpu-baseline --no-family-eval --no-dense-f32 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v $1 B=$2 step %.3f ms'%d['ms_per_step'], ['%.3f'%h['ms'] for h in d['per_hop']])"
  done

done
