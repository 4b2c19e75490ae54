#!/bin/bash
# rocprofv3 PMC passes of one command, one pass per counter group (never combined with the trace domains gpurun refuses).
#   bash tools/pmc.sh <out-name> "<counters of pass 1>" ["<counters of pass 2>" ...] -- python3 <script> [args]
# Writes gpurun_out/<out-name>/<first counter>/... and a per-kernel summary gpurun_out/<out-name>/summary.json
cd /tmp && export TMPDIR=/tmp
NAME=$1; shift
PASSES=()
while [ "$1" != "--" ]; do PASSES+=("$1"); shift; done
shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$NAME
mkdir -p $OUT
for C in "${PASSES[@]}"; do
  tag=$(echo $C | cut -d' ' -f1)
  ( cd $GRAFT_REPO_ROOT && timeout -k 10 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$tag -o p -- "$@" > $OUT/$tag.log 2>&1 ) || echo "pass $tag failed"
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summarise.py $OUT
