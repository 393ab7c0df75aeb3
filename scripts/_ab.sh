python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for I in 1 0; do
  echo "IMPLICIT=$I"; ACM_IMPLICIT=$I python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['eager_ms_per_step'], d['config']['kernel_ms'], d['config']['final_loss'], d['roofline'])"
done
ACM_IMPLICIT=1 ACM_NARROW_PIPE=8192 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
python scripts/bench_configs.py twitch/acmgcnp+A twitch/acmiigcnp squirrel/acmgcnp+A 2>&1 | grep config
