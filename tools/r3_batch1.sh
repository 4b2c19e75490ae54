#!/bin/bash
# round-3 batch: dense tests, full GPU suite, bench line, PMC issue counters of the dense kernel
mkdir -p gpurun_out/r3
python -m pytest tests -m gpu -q -k "dense or split3" > gpurun_out/r3/t_dense.log 2>&1
grep -E "passed|failed|FAILED" gpurun_out/r3/t_dense.log | tail -12
python -m pytest tests -m gpu -x -q > gpurun_out/r3/t_all.log 2>&1
tail -5 gpurun_out/r3/t_all.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-family-eval > gpurun_out/r3/b2.json 2> gpurun_out/r3/b2.err
python - <<EOF2
import json
d=json.loads(open("gpurun_out/r3/b2.json").read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], "dense_f32", d.get("dense_f32"))
print(d["roofline_dense"])
EOF2
bash tools/pmc.sh r3_pmc_dense3 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-family-eval --no-kernel-events --no-dense-f32 > gpurun_out/r3/pmc_dense3.txt 2>&1
grep -A20 "dense_split3" gpurun_out/r3/pmc_dense3.txt | head -40
