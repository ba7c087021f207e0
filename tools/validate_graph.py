#!/usr/bin/env python3
"""Builds the graph of a fuzz_parity configuration (JSON on the command line) and runs the
consistency check of g2s_graph_validate on it (GPU build unless G2S_HOST_BUILD=1)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import cases  # noqa: E402
from gap2seq_amd import lib as product  # noqa: E402

cfg = json.loads(sys.argv[1])
product.load_library()
seqs = cases.toy_genome(cfg["gseed"], cfg["length"], cfg["k"], repeats=cfg["repeats"], tandem=cfg["tandem"],
                        inverted=cfg["inverted"], snp_every=cfg["snp_every"])
g = product.Graph.from_seqs(seqs, cfg["k"], 1)
print("k-mers", g.num_kmers, "unitigs", g.num_unitigs, "violations", g.validate())
if len(sys.argv) > 2:
    g.save(sys.argv[2])
