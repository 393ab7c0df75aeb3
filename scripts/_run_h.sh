cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_oracle.py -q -x -k "acmii_recompute" > gpurun_out/r02h_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02h_pytest.log
python scripts/bench_wide.py --modes default --configs twitch/acmii > gpurun_out/r02h_wide.jsonl 2> gpurun_out/r02h_wide.err
python -m pytest tests/test_gpu_bf16_sweep.py tests/test_gpu_relabel.py -q >> gpurun_out/r02h_pytest.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pq && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d /tmp/pq -o pmc -- python $GRAFT_REPO_ROOT/scripts/bench_wide.py --modes default --configs twitch/acmii > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r02h_pmc.err
python $GRAFT_REPO_ROOT/scripts/rocpd_pmc_summary.py $(find /tmp/pq -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r02h_pmc_sq.csv 2>> $GRAFT_REPO_ROOT/gpurun_out/r02h_pmc.err
