#!/bin/bash
# rocprofv3 kernel stats of a few bench_configs.py configurations -> gpurun_out/configs_kernel_stats.txt
#   bash scripts/configs_kernel_stats.sh [config ...]      (default: the eight below)
cd /tmp; export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/configs_kernel_stats.txt
mkdir -p $REPO/gpurun_out; : > $OUT
CFGS=${@:-twitch/acmiigcnp twitch/acmiigcnp+A twitch/acmgcnpp twitch/acmgcnp+A arxiv-year/acmgcnp arxiv-year/acmsgc-3hop squirrel/acmgcnp+A penn94/acmgcnp/csrX}
for cfg in $CFGS; do
  rm -rf /tmp/pp
  rocprofv3 --kernel-trace --stats -d /tmp/pp -o pp -- python $REPO/scripts/bench_configs.py $cfg > /dev/null 2>&1
  echo "== $cfg   (kernel | calls | avg us | % of GPU time; torch / rocprim preparation kernels omitted)" >> $OUT
  python $REPO/scripts/rocpd_summary.py $(find /tmp/pp -name "*.db" | head -1) | python -c "
import csv, sys, re
rows = list(csv.reader(sys.stdin))
for r in rows[1:]:
    if len(r) < 5 or re.search('at::native|rocprim|hipcub|memory_copies|rocclr', r[0]): continue
    name = re.sub(r'\(anonymous namespace\)::', '', r[0]); name = re.sub(r'^void ', '', name); name = re.sub(r'\(.*', '', name)
    print(f'{name[:60]:60s} {r[1]:>6s} {float(r[3]):9.1f} {r[4]:>6s}')
" | head -14 >> $OUT
done
