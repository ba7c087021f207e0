"""tests/golden/make_readfilter_golden.py — regenerates readfilter_cases.json: a small BAM file (base64) and what the
read filter must extract from it for a handful of calls.

The reference ships no BAM files and cannot be built in this image (GATB, htslib), so the expected texts come from
the repo's Python restatement (oracle/readfilter_ref.py) and are committed ONLY after the product's filter
(gap2seq_amd/csrc/readfilter.cpp, through the C ABI) produced the same bytes: two independent BAM parsers and two
implementations of the passes.  They pin both against silent drift.  Data only: inputs and expected outputs.

Run:  python tests/golden/make_readfilter_golden.py
"""
import base64
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
import bamwriter as BW  # noqa: E402
import readfilter_ref as REF  # noqa: E402
from gap2seq_amd import lib as P  # noqa: E402

CALLS = [
    dict(mean=300, std_dev=20, scaffold="scaf0", breakpoint=700, gap_length=100, flank_length=60),
    dict(mean=300, std_dev=20, scaffold="scaf1", breakpoint=700, gap_length=100),
    dict(mean=300, std_dev=0, scaffold="scaf0", breakpoint=700, gap_length=100, flank_length=10),
    dict(mean=300, std_dev=20, scaffold="scaf0", breakpoint=60, gap_length=5, flank_length=40),
    dict(mean=300, std_dev=20, scaffold="nosuch", breakpoint=700, gap_length=100, flank_length=60),
    dict(mean=300, std_dev=20, scaffold="0", breakpoint=0, gap_length=0, unmapped_only=True),
]


def main():
    refs, recs, _ = BW.simulate_library(20260101, n_scaffolds=2, scaffold_len=1500, gap=(700, 100), read_len=36, mean=300,
                                        sd=20, pairs=40, unmapped_pairs=3, ambiguous=0.05)
    bam = BW.bam_bytes(refs, recs, block=1500)
    out = {"bam_base64": base64.b64encode(bam).decode(), "records": len(recs), "calls": []}
    for kw in CALLS:
        want = REF.read_filter(bam, kw["mean"], kw["std_dev"], kw["scaffold"], kw["breakpoint"], kw.get("gap_length", -1),
                               kw.get("flank_length", -1), kw.get("unmapped_only", False))
        got = P.filter_reads(bam, **kw)
        assert got[:3] == want, kw  # committed only when the two agree
        out["calls"].append({"args": kw, "fasta": want[0], "stdout": want[1], "stderr": want[2]})
    with open(os.path.join(HERE, "readfilter_cases.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("readfilter_cases.json:", len(bam), "BAM bytes,", len(recs), "records,", len(CALLS), "calls,",
          sum(c["fasta"].count(">") for c in out["calls"]), "reads expected in all")


if __name__ == "__main__":
    main()
