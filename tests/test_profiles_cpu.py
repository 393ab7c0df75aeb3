"""The committed evidence under profiles/ stays consistent with the tools that read it:
scripts/make_traffic_json.py must resolve EVERY kernel label of the driver's bench line against the committed counter
CSVs -- a template parameter appended to a kernel may not silently drop its traffic figure again."""
import glob
import json
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import make_traffic_json as M  # noqa: E402


def _rounds():
    out = []
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_fetch_size_kb.csv"))):
        tag = os.path.basename(f).split("_")[0]
        bench = os.path.join(ROOT, "profiles", f"{tag}_bench.json")
        write = os.path.join(ROOT, "profiles", f"{tag}_pmc_write_size_kb.csv")
        if re.fullmatch(r"r\d+", tag) and os.path.exists(bench) and os.path.exists(write) and int(tag[1:]) >= 4:
            out.append((tag, f, write, bench))
    return out


@pytest.mark.parametrize("tag,fetch,write,bench", _rounds(), ids=[r[0] for r in _rounds()])
def test_every_label_of_the_bench_line_resolves_in_the_counter_csvs(tag, fetch, write, bench):
    with open(bench) as fh:
        line = json.loads(fh.read().strip().splitlines()[-1])
    labels = tuple(line["config"]["kernel_ms"])
    assert line["roofline"]["kernel"] in labels
    doc = M.build(M.load(fetch, "FETCH_SIZE_avg"), M.load(write, "WRITE_SIZE_avg"), required=labels)
    for lb in labels:
        assert doc[lb]["hbm_bytes"] > 0 and len(doc[lb]["kernels"]) >= 1, lb
    # a label made of two launches books both (r04's file held the tail kernel only: 27 MB instead of 121 MB)
    if "conv_fwd_tail/F2k3" in doc:
        assert len(doc["conv_fwd_tail/F2k3"]["kernels"]) == 2


def test_needles_survive_an_appended_template_parameter_and_unknown_labels_fail():
    fetch = {"void spmm_narrow_kernel<2, 2, 16, true, EpiBwd, 2, 7>(CsrView, GatherSrc, int)": 10.0,
             "void spmm_narrow_kernel<2, 2, 16, true, EpiRaw, 2>(CsrView)": 20.0,
             "void conv_tail_rows_kernel<2, 2>(acm_conv_fwd_t)": 1.0,
             "(anonymous namespace)::adam_flush_kernel((anonymous namespace)::AdamPack)": 2.0,
             "void (anonymous namespace)::dropout_wide_kernel(long)": 3.0}
    write = {k: 1.0 for k in fetch}
    doc = M.build(fetch, write, required=("conv_bwd_spmm/F2k3", "conv_fwd_tail/F2k3", "adam+flush"))
    assert doc["conv_bwd_spmm/F2k3"]["fetch_kb"] == 10.0 and doc["conv_fwd_tail/F2k3"]["fetch_kb"] == 21.0
    assert "adam" not in doc and "dropout/168114x7" not in doc          # adam_kernel / dropout_kernel: prefixes, not substrings
    with pytest.raises(M.UnresolvedLabel):
        M.build(fetch, write, required=("conv_agg_epi/F64k3i7",))
    with pytest.raises(M.UnresolvedLabel):
        M.build(fetch, write, required=("no_such_label",))
    del fetch["void conv_tail_rows_kernel<2, 2>(acm_conv_fwd_t)"]
    with pytest.raises(M.UnresolvedLabel):                               # half of a two-launch label is not a figure
        M.build(fetch, write, required=("conv_fwd_tail/F2k3",))
