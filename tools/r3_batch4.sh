#!/bin/bash
mkdir -p gpurun_out/r3
python -m pytest tests -m gpu -q -k "gram or train or backward or grad or adam or learn" > gpurun_out/r3/t_train.log 2>&1
grep -E "passed|failed|FAILED" gpurun_out/r3/t_train.log | tail -12
python bench.py --train --steps 10 --warmup 3 > gpurun_out/r3/bench_train2.json 2> gpurun_out/r3/bench_train2.err || tail -5 gpurun_out/r3/bench_train2.err
python - <<EOF2
import json
d=json.loads(open("gpurun_out/r3/bench_train2.json").read().strip().splitlines()[-1])
print("train ms/step %.3f fwd %.2f bwd %.2f other %.2f" % (d["ms_per_step"], d["forward_ms"], d["backward_ms"], d["optimizer_and_host_ms"]))
EOF2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3/prof_train2 -o t -- python3 $GRAFT_REPO_ROOT/bench.py --train --steps 5 --warmup 2 --no-kernel-events > $GRAFT_REPO_ROOT/gpurun_out/r3/prof_train2.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<EOF2
import csv, glob
f = glob.glob("gpurun_out/r3/prof_train2/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print("%6.2f%% %8.3f ms/step x%5d avg %9.1f us  %s" % (float(r["Percentage"]), float(r["TotalDurationNs"]) / 7e6, int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:100]))
print("total kernel ms per step (7 steps)", tot / 7e6)
EOF2
