#!/bin/bash
# Run ON THE GPU BOX (through gpurun): every measurement the committed profiles/ files come from.
#   bash scripts/collect_profiles.sh <tag> [commit]   ->  gpurun_out/profiles_<tag>/
#     bench.json                 python bench.py (default flags: the driver's N=1 line, with cpu_baseline)
#     kernel_stats.csv           rocprofv3 --kernel-trace --stats of the same command (per-kernel time)
#     step_timeline.txt          ordered kernels of the last graph-replayed steps
#     pmc_fetch_size_kb.csv / pmc_write_size_kb.csv / pmc_cache.csv   separate --pmc passes (never with a trace domain)
#     pmc_traffic.json           HBM bytes per launch of the step's kernels (2 * FETCH_SIZE + WRITE_SIZE, KiB -> B)
set -u
TAG=${1:-run}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
COMMIT=${2:-unknown}
BENCH="python $REPO/bench.py"
$BENCH > $OUT/bench.json 2> $OUT/bench.err
rm -rf /tmp/prof_kt && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- $BENCH --no-cpu-baseline --no-extras --no-check > /dev/null 2> $OUT/kt.err
DB=$(find /tmp/prof_kt -name "*.db" | head -1)
python $REPO/scripts/rocpd_summary.py $DB > $OUT/kernel_stats.csv
python $REPO/scripts/rocpd_sequence.py $DB 60 > $OUT/step_timeline.txt
for PASS in "FETCH_SIZE:pmc_fetch_size_kb" "WRITE_SIZE:pmc_write_size_kb" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum:pmc_cache"; do
  CNT=${PASS%%:*}; NAME=${PASS##*:}
  rm -rf /tmp/prof_pmc && timeout 400 rocprofv3 --pmc $CNT --kernel-trace -d /tmp/prof_pmc -o pmc -- python $REPO/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-check > /dev/null 2> $OUT/$NAME.err
  DB=$(find /tmp/prof_pmc -name "*.db" | head -1)
  python $REPO/scripts/rocpd_pmc_summary.py $DB > $OUT/$NAME.csv 2>> $OUT/$NAME.err
done
if [ "$COMMIT" = "unknown" ]; then echo "collect_profiles.sh: pass the commit the box runs (git rev-parse --short HEAD) as the second argument" >&2; fi
# every label of the bench line must resolve against the counter CSVs (exit status 1 and a message otherwise)
python $REPO/scripts/make_traffic_json.py $OUT/pmc_fetch_size_kb.csv $OUT/pmc_write_size_kb.csv $COMMIT --require-from $OUT/bench.json > $OUT/pmc_traffic.json || echo "collect_profiles.sh: pmc_traffic.json is INCOMPLETE (see the message above)" >&2
# the driver's line LAST, quoting the counters just collected (same kernel sources by construction: no stale traffic)
mv $OUT/bench.json $OUT/bench_first.json
$BENCH --traffic-json $OUT/pmc_traffic.json > $OUT/bench.json 2>> $OUT/bench.err
tail -1 $OUT/bench.json | cut -c1-400
