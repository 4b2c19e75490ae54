for v in 2 18 34; do
  export RG_BENCH_FORCE_WALK=$v
  bash tools/pmc.sh r2_pmc_v$v "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU" -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-family-eval --no-kernel-events > gpurun_out/r2_pmc_v$v.txt 2>&1
  grep -A14 "layer_fwd_wp" gpurun_out/r2_pmc_v$v.txt | grep "wp_kernel\|dur_ns\|SQ_"
done
