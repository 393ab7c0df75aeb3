#!/bin/bash
# Round 6, VERDICT r05 items 3 + 4: name the bounds with counters.
#   (a) the counters this box offers for the texture path (TA / TCP / TD), so that the passes below use names that exist
#   (b) SQ / occupancy / cache / LDS passes of the dominant kernel agg_bwd16_gather
#   (c) TA / TCP passes of the two output-layer narrow gathers
# Separate --pmc passes, --kernel-trace only (scripts/pmc_kernel.sh).  Output: gpurun_out/r06_*.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
(cd /tmp && timeout 120 rocprofv3 -L 2>&1 | grep -oE "\b(TA|TCP|TD|TCC|SQ|GRBM)_[A-Za-z0-9_]+" | sort -u > /root/repo/gpurun_out/r06_counters_avail.txt)
bash scripts/pmc_kernel.sh r06_bwd16 "agg_bwd16_gather|agg_epi16" "sq sq2 occ cache lds mfma"
bash scripts/pmc_kernel.sh r06_narrow "spmm_narrow_kernel<2, 2|conv_tail|agg_bwd16_gather" "ta1 ta2 ta3 tcp1 tcp2 tcp3 tcp4 sq sq2 cache"
