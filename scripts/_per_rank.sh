cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python scripts/shard8_per_rank.py --dataset arxiv-year --method acmsgc --hops 3 --world 1,2,4,8 --ranks 0 > $O/r05_shard_per_rank_arxiv.json 2> $O/r05_shard_arxiv.err
timeout 900 python scripts/shard8_per_rank.py --dataset penn94 --method acmsgc --hops 3 --world 1,2,4,8 --ranks 0 > $O/r05_shard_per_rank_penn94.json 2> $O/r05_shard_penn94.err
timeout 1200 python scripts/shard8_per_rank.py --dataset pokec --method acmgcnp --world 1,2,4,8 --ranks 0 --steps 3 > $O/r05_shard_per_rank_pokec.json 2> $O/r05_shard_pokec.err
timeout 1500 python -m pytest tests/test_gpu_scale.py tests/test_gpu_sharded.py -q 2>&1 | tail -5 > $O/r05_scale_tests.log
cat $O/r05_scale_tests.log
