for P in 0 4096 8192 16384; do
  echo "PIPE=$P"; ACM_NARROW_PIPE=$P python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['eager_ms_per_step'], d['config']['kernel_ms'])"
done
ACM_NARROW_PIPE=0 python scripts/probe_gather.py 2>&1 | grep -E "real|random"
ACM_NARROW_PIPE=8192 python scripts/probe_gather.py 2>&1 | grep -E "real|random"
