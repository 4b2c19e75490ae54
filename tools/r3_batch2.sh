#!/bin/bash
mkdir -p gpurun_out/r3
python -m pytest tests -m gpu -q -k "dense or split3 or extrap" > gpurun_out/r3/t_sel.log 2>&1
grep -E "passed|failed|FAILED" gpurun_out/r3/t_sel.log | tail -12
python -m pytest tests -m gpu -x -q > gpurun_out/r3/t_all.log 2>&1
tail -4 gpurun_out/r3/t_all.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r3/bench_default.json 2> gpurun_out/r3/bench_default.err || tail -5 gpurun_out/r3/bench_default.err
python bench.py --train --steps 10 --warmup 3 > gpurun_out/r3/bench_train.json 2> gpurun_out/r3/bench_train.err || tail -5 gpurun_out/r3/bench_train.err
python bench.py --config X --steps 10 --warmup 2 > gpurun_out/r3/bench_X.json 2> gpurun_out/r3/bench_X.err || tail -5 gpurun_out/r3/bench_X.err
python bench.py --config X --train --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3/bench_X_train.json 2> gpurun_out/r3/bench_X_train.err || tail -5 gpurun_out/r3/bench_X_train.err
python - <<EOF2
import json
for f in ("bench_default","bench_train","bench_X","bench_X_train"):
    try:
        d=json.loads(open("gpurun_out/r3/%s.json"%f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "no json", e); continue
    print(f, "ms/step %.3f value %.4g dtype %s" % (d["ms_per_step"], d["value"], d["dtype"]))
    r=d.get("roofline") or {}
    print("   roofline", r.get("bound"), r.get("frac"), (r.get("hbm_algorithmic") or {}).get("frac"))
    for k in ("forward_ms","backward_ms","optimizer_and_host_ms","dense_f32","parity"):
        if k in d: print("   ",k,d[k])
    if d.get("family_eval"): print("    family", {k:v for k,v in d["family_eval"].items() if k!="path"})
    if d.get("roofline_dense"): print("    dense", {k:d["roofline_dense"][k] for k in ("kernel","bound","frac","avg_launch_ms","hbm_algorithmic_frac")})
    if d.get("per_hop"): print("    per_hop", [(h["hop"], round(h["ms"],3), round(h["l2_gather_frac"],3)) for h in d["per_hop"]])
    if d.get("cpu_baseline"): print("    cpu", d["cpu_baseline"])
EOF2
bash tools/pmc.sh r3_pmc_dense3b "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-family-eval --no-kernel-events --no-dense-f32 > gpurun_out/r3/pmc_dense3b.txt 2>&1
grep -A22 "dense_split3" gpurun_out/r3/pmc_dense3b.txt | grep -E "dur_ns|BANK|IDX_ACTIVE|MFMA_BUSY|WAIT_ANY|WAVE_CYCLES|INST_ANY|INSTS_LDS"
