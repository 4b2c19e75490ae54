#!/bin/bash
# dense tests + one bench line (round-3 iteration helper)
mkdir -p gpurun_out/r3
python -m pytest tests -m gpu -x -q -k "dense or split3" > gpurun_out/r3/t_dense.log 2>&1
tail -15 gpurun_out/r3/t_dense.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-family-eval > gpurun_out/r3/b1.json 2> gpurun_out/r3/b1.err
tail -c 600 gpurun_out/r3/b1.err
python - <<EOF2
import json
d=json.loads(open("gpurun_out/r3/b1.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d.get("dense_f32"))
print(d["roofline_dense"])
print(d["per_hop"])
EOF2
