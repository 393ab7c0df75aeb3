#!/usr/bin/env python3
"""HBM bytes per launch of the bench step's kernels from the FETCH_SIZE / WRITE_SIZE PMC summaries
(scripts/rocpd_pmc_summary.py CSVs):  hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, FETCH_SIZE doubled as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950.  Keys are bench.py's kernel labels; a label
made of several launches (gather + row epilogue) sums them.

    python scripts/make_traffic_json.py pmc_fetch_size_kb.csv pmc_write_size_kb.csv [commit] > pmc_traffic.json

The document is stamped with ``_kernel_source_hash`` (sha256 over acm_gnn_amd/csrc and include/, as they were when the
counters were collected): bench.py attaches the traffic figure only while the kernels it runs hash to the same value,
so a stale file cannot be quoted against changed kernels.
"""
import csv
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALTERNATIVES = {"conv_agg_bwd/F64k3i7", "conv_agg_epi/F64k3i7"}     # labels whose needles are alternatives, not a sum


def kernel_source_hash(root=ROOT):
    """sha256 over the kernel sources and the ABI header (sorted by name)."""
    h = hashlib.sha256()
    files = [os.path.join(root, "include", "acm_hip.h")]
    csrc = os.path.join(root, "acm_gnn_amd", "csrc")
    files += sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h", ".cpp")))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]

# bench label -> substrings identifying its kernels in the profiler's names
LABELS = {
    "conv_agg_fwd/F64k3i7": ["agg_fused_pair_kernel", "agg_fused_kernel<8"],
    "conv_agg_bwd/F64k3i7": ["agg_bwd16_kernel", "agg_bwd_kernel<8, 3"],     # (whichever ran: the first needle found wins)
    "conv_agg_epi/F64k3i7": ["agg_epi16_kernel", "agg_epilogue_kernel<8, 3>"],   # pipelined step: the row-local stage of the forward alone
    "conv_agg_bwd+gather/F64k3i7": ["agg_bwd16_gather_kernel<3, true, false>"],        # ... and the backward carrying the next step's gather
    "conv_agg_bwd+gather+proj/F64k3i7": ["agg_bwd16_gather_kernel<3, true, true>"],   # ... and the output layer's projection backward
    "conv_agg_bwd+proj/F64k3i7": ["agg_bwd16_kernel<3, 8, true, true, true>"],
    "conv_fwd_tail/F2k3": ["spmm_narrow_kernel<2, 2, 16, true, EpiRaw>", "conv_tail_rows_kernel<2, 2>"],   # + loss + K3
    "conv_bwd_spmm/F2k3": ["spmm_narrow_kernel<2, 2, 16, true, EpiBwd>"],
    "proj_bwd/168114x64x6": ["proj_bwd_kernel<6>"],
    "proj_fwd/168114x64x6": ["proj_fwd_kernel<2>"],
    "dropout/168114x7": ["dropout_kernel"],
    "dropout/168120x7": ["dropout_kernel"],
    "reduce_flush": ["reduce_segments_kernel"],          # every deferred second phase of the step, one launch
    "adam": ["adam_kernel"],
    "adam+flush": ["adam_flush_kernel"],                 # the update launch that also runs the deferred second phases
}


def load(path, column):
    out = {}
    with open(path) as fh:
        for row in csv.DictReader(fh):
            out[row["kernel"]] = float(row[column])
    return out


def main(fetch_csv, write_csv, commit=None):
    fetch, write = load(fetch_csv, "FETCH_SIZE_avg"), load(write_csv, "WRITE_SIZE_avg")
    doc = {"_commit": commit or "unknown", "_kernel_source_hash": kernel_source_hash(),
           "_doc": "HBM-side bytes per launch from rocprofv3 PMC (separate --pmc FETCH_SIZE / WRITE_SIZE passes over "
                   "bench.py's default workload). hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE doubled as "
                   "MI355X_MICROARCH.md prescribes for gfx950 (calibrated there on wide coalesced streams; the random "
                   "16-32 B gathers here are outside that calibration, so read ratios to algorithmic bytes as 1x..2x)."}
    for label, needles in LABELS.items():
        f = w = 0.0
        found = []
        for needle in needles:
            for name in fetch:
                if needle in name:
                    f += fetch[name]
                    w += write.get(name, 0.0)
                    found.append(needle)
                    break
            if found and label in ALTERNATIVES:
                break
        if found:
            doc[label] = {"fetch_kb": round(f, 1), "write_kb": round(w, 1), "kernels": found,
                          "hbm_bytes": int((2 * f + w) * 1024)}
    json.dump(doc, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
