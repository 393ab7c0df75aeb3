#!/bin/bash
# Run ON THE GPU BOX: L1 -> L2 read requests of the output-layer gathers with and without the hub rows in LDS
# (acm_tuning_t.gather_forms bit 2), one rocprofv3 --pmc pass each over bench.py's default workload.
#   bash scripts/pmc_hub_ab.sh  ->  gpurun_out/pmc_hub_{on,off}.csv
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
for MODE in on off; do
  if [ $MODE = off ]; then export ACM_TUNING="gather_forms=1"; else unset ACM_TUNING; fi
  rm -rf /tmp/prof_hub && timeout 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace -d /tmp/prof_hub -o pmc -- python $REPO/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-check > /dev/null 2> $OUT/pmc_hub_$MODE.err
  DB=$(find /tmp/prof_hub -name "*.db" | head -1)
  python $REPO/scripts/rocpd_pmc_summary.py $DB > $OUT/pmc_hub_$MODE.csv 2>> $OUT/pmc_hub_$MODE.err
done
