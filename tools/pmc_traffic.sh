#!/bin/bash
# HBM traffic of the layer_fwd launches of one eval step (profiles/traffic_layer_fwd.json entries): separate --pmc passes as
# MI355X_MICROARCH.md prescribes (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2: never in one pass).
#   bash tools/pmc_traffic.sh <config> <batch>
CFG=${1:-C2}; B=${2:-1024}
bash tools/pmc.sh r3_traffic_${CFG}_${B} "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" -- python3 bench.py --config $CFG --batch $B --steps 2 --warmup 1 --no-cpu-baseline --no-family-eval --no-kernel-events --no-dense-f32 > gpurun_out/r3_traffic_${CFG}_${B}.txt 2>&1
tail -50 gpurun_out/r3_traffic_${CFG}_${B}.txt
