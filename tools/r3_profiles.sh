#!/bin/bash
# round-3 evidence run: bench lines of every configuration, rocprofv3 kernel stats, PMC traffic, the N = 2 rehearsal
O=gpurun_out/r3p; mkdir -p $O
python bench.py --steps 100 --warmup 5 > $O/bench_default_steps100.json 2> $O/bench_default.err || tail -3 $O/bench_default.err
for c in "C2 64" "C2 256" "C3 64" "C3 256" "C4 64" "C4 256"; do set -- $c
  python bench.py --config $1 --batch $2 --steps 10 --warmup 3 --no-family-eval > $O/bench_$1_B$2.json 2>/dev/null; done
python bench.py --config C5 --batch 64 --steps 10 --warmup 3 > $O/bench_C5_B64.json 2>/dev/null
python bench.py --config X --steps 10 --warmup 2 > $O/bench_X_B64.json 2>/dev/null
python bench.py --config X --train --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_X_B64_train.json 2>/dev/null
python bench.py --train --steps 10 --warmup 3 > $O/bench_train_C2_B256.json 2>/dev/null
echo "bench lines done"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_bench -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-family-eval --no-dense-f32 > $GRAFT_REPO_ROOT/$O/bench_line_of_the_kernel_stats_run.json 2>/dev/null )
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_train -o t -- python3 $GRAFT_REPO_ROOT/bench.py --train --steps 5 --warmup 2 --no-kernel-events > $GRAFT_REPO_ROOT/$O/bench_line_of_the_train_stats_run.json 2>/dev/null )
cp $O/prof_bench/b_kernel_stats.csv $O/kernel_stats_bench_steps10_warmup2.csv 2>/dev/null || find $O/prof_bench -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_bench_steps10_warmup2.csv \;
find $O/prof_train -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_train_C2_B256_steps5_warmup2.csv \;
rm -rf $O/prof_bench $O/prof_train
echo "rocprof stats done"
bash tools/pmc_traffic.sh C2 1024 > /dev/null 2>&1
cp gpurun_out/r3_traffic_C2_1024/summary.json $O/pmc_traffic_C2_B1024_v10_summary.json
echo "pmc done"
for extra in "" "--eval-collective allreduce" "--global-batch 256 --config C4" "--train --batch 128"; do
  tag=$(echo "$extra" | tr -d ' -' ); tag=${tag:-allgather}
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --steps 3 --warmup 1 --batch 256 --backend gloo --no-cpu-baseline --no-family-eval $extra > $O/bench_2rank_gloo_rehearsal_$tag.json 2> $O/rehearsal_$tag.err || tail -3 $O/rehearsal_$tag.err
done
python - <<EOF2
import json, glob, os
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d.get("roofline") or {}
        print("%-48s %8.3f ms/step %10.4g %s n=%d roof %s %.3f" % (os.path.basename(f), d["ms_per_step"], d["value"], d["scaling"], d["n_gpus"], r.get("bound"), r.get("frac") or -1))
    except Exception as e:
        print(os.path.basename(f), "BROKEN", e)
EOF2
