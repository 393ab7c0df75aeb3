#!/bin/bash
# Counter passes for the kernels of the benchmark step whose name contains <kernel-substring>:
#   scripts/pmc_kernel.sh <out-prefix> <kernel-substring> "<pass> <pass> ..." [env assignments...]
# PMC_CMD (environment) replaces the profiled command (default: bench.py's short run), e.g.
#   PMC_CMD="python scripts/bench_configs.py twitch/acmiigcnp" scripts/pmc_kernel.sh acmii_v acmii_v "sq sq2 mfma"
# passes: sq sq2 cache ta lds fetch write.  Writes gpurun_out/<prefix>_<pass>.csv.  Separate --pmc passes, --kernel-trace only
# (never with another trace domain), each under its own timeout.
set -u
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
PREFIX=$1; MATCH=$2; PASSES=$3; shift 3
OUT=$REPO/gpurun_out
mkdir -p $OUT
for NAME in $PASSES; do
  case $NAME in
    sq) CNT="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY";;
    sq2) CNT="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM";;
    cache) CNT="TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum";;
    ta) CNT="TA_BUSY_avr TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum";;
    lds) CNT="SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM_WR";;
    mfma) CNT="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY";;
    # (round 6) the texture path, two counters of a block per pass: r05's six-counter "ta" pass returned no rows
    ta1) CNT="TA_TA_BUSY_sum TA_BUSY_avr";;
    ta2) CNT="TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum";;
    ta3) CNT="TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum";;
    tcp1) CNT="TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum";;
    tcp2) CNT="TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum";;
    tcp3) CNT="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum";;
    tcp4) CNT="TCP_GATE_EN1_sum TCP_GATE_EN2_sum";;
    occ) CNT="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LEVEL_WAVES GRBM_GUI_ACTIVE";;
    fetch) CNT="FETCH_SIZE";;
    write) CNT="WRITE_SIZE";;
  esac
  rm -rf /tmp/prof_pmc
  CMD=${PMC_CMD:-"python $REPO/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-check"}
  (cd $REPO && env "$@" timeout 300 rocprofv3 --pmc $CNT --kernel-trace -d /tmp/prof_pmc -o pmc -- $CMD > /dev/null 2> $OUT/${PREFIX}_$NAME.err)
  DB=$(find /tmp/prof_pmc -name "*.db" | head -1)
  [ -n "$DB" ] && python $REPO/scripts/rocpd_pmc_summary.py $DB 2>> $OUT/${PREFIX}_$NAME.err | grep -E "^kernel|$MATCH" > $OUT/${PREFIX}_$NAME.csv
done
