#!/bin/bash
# rocprofv3 kernel stats of one bench configuration: bash tools/r3_prof_cfg.sh <tag> <bench.py args...>  -> gpurun_out/r3/kernel_stats_<tag>.csv
TAG=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $O
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o p -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $O/bench_line_$TAG.json 2>$O/bench_line_$TAG.err )
find $O/prof_$TAG -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$TAG.csv \;
rm -rf $O/prof_$TAG
python3 - <<EOF2
import csv
rows=list(csv.DictReader(open("$O/kernel_stats_$TAG.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:22]:
    print('%-64s calls %5s avg %9.1f us  %5.1f%%'%(r['Name'][:64], r['Calls'], float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
EOF2
