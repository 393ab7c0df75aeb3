cd $GRAFT_REPO_ROOT
for mode in default vec; do
  if [ $mode = vec ]; then export ACM_WIDE_VEC=1; else unset ACM_WIDE_VEC; fi
  echo "== $mode"
  python scripts/bench_configs.py squirrel/acmgcnp+A/csrX chameleon/acmgcnp+A cora/acmgcn/csrX 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['config'], d['graph_ms'], {k:v for k,v in list(d['kernel_us'].items())[:6]})"
done
