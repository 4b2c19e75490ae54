#!/bin/bash
mkdir -p gpurun_out/r3
python -m pytest tests -m gpu -q -k "extrap" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3/prof_train -o t -- python3 $GRAFT_REPO_ROOT/bench.py --train --steps 5 --warmup 2 --no-kernel-events > $GRAFT_REPO_ROOT/gpurun_out/r3/prof_train.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<EOF2
import csv, glob
f = glob.glob("gpurun_out/r3/prof_train/**/*kernel_stats.csv", recursive=True)
print(f)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    print("%6.2f%% %8.3f ms/step x%5d avg %9.1f us  %s" % (float(r["Percentage"]), float(r["TotalDurationNs"]) / 7e6, int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:110]))
print("total kernel ms per step (7 steps)", tot / 7e6)
EOF2
for c in "C3 256" "C4 256" "C4 64" "C3 64" "C2 256" "C2 64"; do set -- $c; python bench.py --config $1 --batch $2 --steps 10 --warmup 3 --no-cpu-baseline --no-family-eval > gpurun_out/r3/bench_$1_B$2.json 2>/dev/null; python - <<EOF3
import json
d=json.loads(open("gpurun_out/r3/bench_$1_B$2.json").read().strip().splitlines()[-1])
print("$1 B=$2: %.2f ms/step, %.3g edges/s, layer l2 frac %.2f, dense %s %.2f ms/launch, f32 step %s" % (d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline_dense"]["kernel"], d["roofline_dense"]["avg_launch_ms"], d["dense_f32"] and round(d["dense_f32"]["ms_per_step"],2)))
EOF3
done
