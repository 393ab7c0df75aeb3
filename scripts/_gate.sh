cd $GRAFT_REPO_ROOT
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/gate_r05.log
tail -8 gpurun_out/gate_r05.log
for f in accuracy_replay_cora accuracy_replay_film_v0 accuracy_replay_film_v1 accuracy_replay_squirrel accuracy_bf16_film_v1 accuracy_bf16_squirrel fullsize_parity bf16_sweep; do ls -la gpurun_out/$f.json 2>/dev/null | cut -c30-; done
