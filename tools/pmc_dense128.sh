cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc128b
mkdir -p $OUT
for C in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
  tag=$(echo $C | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$tag -o p -- python3 $GRAFT_REPO_ROOT/bench.py --config C4 --batch 64 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > $OUT/$tag.log 2>&1 || echo "failed $tag"
done
ls $OUT
