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

# bench label -> needles identifying its kernels in the profiler's names.  A needle is the kernel's base name plus the
# LEADING template arguments that tell the instantiations of one step apart, WITHOUT the closing bracket, so a template
# parameter appended later (round 4: `spmm_narrow_kernel<…, EpiRaw>` became `<…, EpiRaw, 2>` and two labels silently
# vanished from r04_pmc_traffic.json) still matches.  `match()` anchors the needle at the start of the base name.
LABELS = {
    "conv_agg_fwd/F64k3i7": ["agg_fused_pair_kernel", "agg_fused_kernel<8"],
    "conv_agg_bwd/F64k3i7": ["agg_bwd16_kernel<3, 8", "agg_bwd_kernel<8, 3"],     # (whichever ran: the first needle found wins)
    "conv_agg_epi/F64k3i7": ["agg_epi16_kernel<3, 8", "agg_epilogue_kernel<8, 3"],   # pipelined step: the row-local stage of the forward alone
    "conv_agg_bwd+gather/F64k3i7": ["agg_bwd16_gather_kernel<3, true, false"],        # ... and the backward carrying the next step's gather
    "conv_agg_bwd+gather+proj/F64k3i7": ["agg_bwd16_gather_kernel<3, true, true"],   # ... and the output layer's projection backward
    "conv_agg_bwd+proj/F64k3i7": ["agg_bwd16_kernel<3, 8, true, true, true"],
    "conv_fwd_tail/F2k3": ["spmm_narrow_kernel<2, 2, 16, true, EpiRaw", "conv_tail_rows_kernel<2, 2"],   # gather + (head + loss + K3)
    "conv_bwd_spmm/F2k3": ["spmm_narrow_kernel<2, 2, 16, true, EpiBwd"],
    "proj_bwd/168114x64x6": ["proj_bwd_kernel<6"],
    "proj_fwd/168114x64x6": ["proj_fwd_kernel<2"],
    "dropout/168114x7": ["dropout_kernel"],
    "dropout/168120x7": ["dropout_kernel"],
    "reduce_flush": ["reduce_segments_kernel"],          # every deferred second phase of the step, one launch
    "adam": ["adam_kernel"],
    "adam+flush": ["adam_flush_kernel"],                 # the update launch that also runs the deferred second phases
}


def base_name(kernel):
    """The profiler's kernel name without return type, anonymous namespace and argument list."""
    name = kernel.replace("(anonymous namespace)::", "")
    name = name[5:] if name.startswith("void ") else name
    depth = 0
    for i, ch in enumerate(name):           # cut at the '(' of the argument list (template arguments hold no parentheses here)
        depth += ch == "<"
        depth -= ch == ">"
        if ch == "(" and depth == 0:
            return name[:i]
    return name


def match(needle, kernel):
    """`needle` is a prefix of the kernel's base name that ends at a template-argument boundary."""
    name = base_name(kernel)
    if not name.startswith(needle):
        return False
    rest = name[len(needle):]
    return rest == "" or rest[0] in "<,>" or needle.endswith("<")


class UnresolvedLabel(KeyError):
    pass


def build(fetch, write, required=()):
    """label -> {fetch_kb, write_kb, kernels, hbm_bytes}.  A label whose needles are a SUM needs every needle (a gather
    booked without its row kernel is a wrong number, not a partial one); a label listed in `required` that does not
    resolve raises UnresolvedLabel — bench.py's labels must never silently lose their traffic figure."""
    doc = {}
    for label, needles in LABELS.items():
        f = w = 0.0
        found = []
        for needle in needles:
            hits = [name for name in fetch if match(needle, name)]
            if hits:
                f += fetch[hits[0]]
                w += write.get(hits[0], 0.0)
                found.append(needle)
                if label in ALTERNATIVES:
                    break
        complete = bool(found) and (label in ALTERNATIVES or len(found) == len(needles))
        if complete:
            doc[label] = {"fetch_kb": round(f, 1), "write_kb": round(w, 1), "kernels": found,
                          "hbm_bytes": int((2 * f + w) * 1024)}
        elif label in required:
            raise UnresolvedLabel(f"{label}: needles {needles} matched {found or 'nothing'} among the profiled kernels")
    for label in required:
        if label not in LABELS:
            raise UnresolvedLabel(f"{label}: no entry in LABELS")
    return doc


def load(path, column):
    out = {}
    with open(path) as fh:
        for row in csv.DictReader(fh):
            out[row["kernel"]] = float(row[column])
    return out


def main(fetch_csv, write_csv, commit=None, required=()):
    fetch, write = load(fetch_csv, "FETCH_SIZE_avg"), load(write_csv, "WRITE_SIZE_avg")
    doc = {"_commit": commit or "unknown", "_kernel_source_hash": kernel_source_hash(),
           "_doc": "HBM-side bytes per launch from rocprofv3 PMC (separate --pmc FETCH_SIZE / WRITE_SIZE passes over "
                   "bench.py's default workload). hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE doubled as "
                   "MI355X_MICROARCH.md prescribes for gfx950 (calibrated there on wide coalesced streams; the random "
                   "16-32 B gathers here are outside that calibration, so read ratios to algorithmic bytes as 1x..2x)."}
    doc.update(build(fetch, write, required))
    json.dump(doc, sys.stdout, indent=1)


if __name__ == "__main__":
    # make_traffic_json.py fetch.csv write.csv [commit] [--require label,label,...] [--require-from bench.json]
    argv = sys.argv[1:]
    req = ()
    if "--require" in argv:
        i = argv.index("--require")
        req = tuple(x for x in argv[i + 1].split(",") if x)
        del argv[i:i + 2]
    if "--require-from" in argv:                 # ... or read off a bench line (config.kernel_ms of bench.py's JSON)
        i = argv.index("--require-from")
        with open(argv[i + 1]) as fh:
            req = req + tuple(json.loads(fh.read().strip().splitlines()[-1])["config"]["kernel_ms"])
        del argv[i:i + 2]
    main(argv[0], argv[1], argv[2] if len(argv) > 2 else None, req)
