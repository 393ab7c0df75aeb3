cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_oracle.py -q -x -k "kernel_forms or aggregate_first" > gpurun_out/r02s_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02s_pytest.log
for m in 0 1; do
  if [ $m = 1 ]; then export ACM_AGG_MFMA=1; else unset ACM_AGG_MFMA; fi
  python bench.py --no-cpu-baseline --no-extras --no-check 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('MFMA=$m', d['ms_per_step'], d['config']['kernel_ms'])"
done > gpurun_out/r02s_ab.txt 2>&1
