#!/bin/bash
# Counter passes (separate --pmc runs, kernel trace only) over the split-bf16 projections (scripts/probe_gemm_pmc.py) and the
# arXiv-year-shaped ACM-GCN+ step (for the literal layer's row-local backward):  gpurun_out/pmc_gemm_<pass>.csv
set -u
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
for NAME in sq lds mfma fetch write; do
  case $NAME in
    sq) CNT="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY";;
    lds) CNT="SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CU_CYCLES";;
    mfma) CNT="SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES";;
    fetch) CNT="FETCH_SIZE";;
    write) CNT="WRITE_SIZE";;
  esac
  for WHAT in gemm arxiv; do
    rm -rf /tmp/prof_pmc
    if [ $WHAT = gemm ]; then CMD="python $REPO/scripts/probe_gemm_pmc.py"; MATCH="gemm_bx3|splitk"; else CMD="python $REPO/scripts/bench_configs.py arxiv-year/acmgcnp"; MATCH="bwd_local16|gemm_bx3"; fi
    timeout 300 rocprofv3 --pmc $CNT --kernel-trace -d /tmp/prof_pmc -o pmc -- $CMD > /dev/null 2> $OUT/pmc_${WHAT}_$NAME.err
    DB=$(find /tmp/prof_pmc -name "*.db" | head -1)
    [ -n "$DB" ] && python $REPO/scripts/rocpd_pmc_summary.py $DB 2>> $OUT/pmc_${WHAT}_$NAME.err | grep -E "^kernel|$MATCH" > $OUT/pmc_${WHAT}_$NAME.csv
  done
done
true
