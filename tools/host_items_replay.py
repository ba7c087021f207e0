#!/usr/bin/env python3
"""Replay, without a GPU, the host's share of a list in resident mode: the closures the device handed over
(G2S_HOST_ITEMS_DUMP=<file> on a bench.py run of the same config) through g2s_test_post_segments — the same
seg_analyze / seg_traceback the batch path calls.  Prints the time per item and writes a digest of every result,
so that two builds of the library can be compared (G2S_LIBRARY=<other .so>).

  python tools/host_items_replay.py gpurun_out/r04host/items_c5.bin --config C5 [--digest out.json] [--repeat 5]
"""
import argparse, ctypes as C, json, os, struct, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
import gap2seq_amd as P
from gap2seq_amd import lib as L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dump")
    ap.add_argument("--config", default="C5")
    ap.add_argument("--variant", type=int, default=3)
    ap.add_argument("--fuz", type=int, default=10)
    ap.add_argument("--digest", default="")
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--top", type=int, default=8)
    a = ap.parse_args()
    genome_bp, k, ngaps, min_len, max_len, d_err, _ = bench.CONFIGS[a.config]
    reads = P.G2S.synth_genome(genome_bp, a.variant, bench.GENOME_SEED)
    seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
    gaps = bench.parse_gaps(P.G2S.synth_gaps(reads, k, a.fuz, ngaps, min_len, max_len, bench.GAP_SEED), a.fuz)
    graph = P.Graph.from_seqs(seqs, k, 1)
    params = L.make_params(d_err=d_err)
    lib = L.load_library()
    raw = open(a.dump, "rb").read()
    items, at = [], 0
    while at < len(raw):
        hd = struct.unpack_from("<8i", raw, at); at += 32
        segs = np.frombuffer(raw, dtype=np.uint32, count=hd[1] * 8, offset=at).copy(); at += hd[1] * 32
        items.append((hd, segs))
    print("%d items, %d segments, largest %d" % (len(items), sum(h[1] for h, _ in items), max(h[1] for h, _ in items)))
    digest, times = [], []
    for hd, segs in items:
        g = gaps[hd[0]]
        gap = L.Gap(g["left"], g["right"], g["gap_len"], g["lmf"], g["rmf"])
        arr, keep = L._gap_array([gap])
        lens = (C.c_int32 * 2)(hd[4], hd[5])
        res = L.g2s_result()
        on = C.c_int32(0)
        buf = C.create_string_buffer(g["gap_len"] + k + d_err + g["lmf"] + g["rmf"] + 3)
        best = 1e9
        for _ in range(a.repeat):
            t0 = time.perf_counter()
            rc = lib.g2s_test_post_segments(graph.h, C.byref(params), arr, hd[1], segs.ctypes.data_as(C.POINTER(C.c_uint32)), hd[2], hd[3],
                                            lens, hd[6], hd[7], 1, 0, C.byref(res), buf, C.byref(on))
            best = min(best, time.perf_counter() - t0)
            assert rc == 0, (rc, L.last_error() if hasattr(L, "last_error") else "")
        times.append(best)
        digest.append(dict(gap=hd[0], n_segs=hd[1], on=on.value, count=res.count, draws=res.draws, flags=res.flags,
                           sub=[res.vertices, res.edges, res.nontrivial_components, res.size_nontrivial_components,
                                res.vertices_final, res.edges_final], fuz=[res.left_fuz, res.right_fuz],
                           crc=zlib.crc32(buf.raw.split(b"\0")[0] if res.left_fuz == g["lmf"] else buf.raw)))
    order = sorted(range(len(items)), key=lambda i: -times[i])
    print("sum %.2f ms, max %.3f ms (hook: analysis + three tracebacks + flank lookup)" % (1e3 * sum(times), 1e3 * max(times)))
    for i in order[:a.top]:
        print("  gap %4d: %5d segments  %.3f ms" % (items[i][0][0], items[i][0][1], 1e3 * times[i]))
    if a.digest:
        json.dump(digest, open(a.digest, "w"))
        print("digest ->", a.digest)


if __name__ == "__main__":
    main()
