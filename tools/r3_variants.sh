#!/bin/bash
mkdir -p gpurun_out/r3
for v in "$@"; do
  export RG_LIB=$PWD/red-gnn_amd/libredgnn_$v.so
  python tools/r3_debug2.py 2>/dev/null | tail -1
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-family-eval --no-dense-f32 > gpurun_out/r3/bv_$v.json 2>/dev/null
  python - <<EOF2
import json
d=json.loads(open("gpurun_out/r3/bv_$v.json").read().strip().splitlines()[-1])
print("$v", "ms/step", round(d["ms_per_step"],3), "dense avg launch ms", round(d["roofline_dense"]["avg_launch_ms"],3))
EOF2
done
