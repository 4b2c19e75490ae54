#!/bin/bash
# A/B of library builds on the default bench step: bash tools/r3_ab_bench.sh <variant names...> (red-gnn_amd/libredgnn_<name>.so)
for v in "$@"; do
  RG_LIB=$PWD/red-gnn_amd/libredgnn_$v.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-family-eval --no-dense-f32 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v step %.3f ms'%d['ms_per_step'], ['%.3f'%h['ms'] for h in d['per_hop']])"
done
