"""GPU box: one long list as K sub-lists in flight (g2s_fill_begin / g2s_fill_end) against one g2s_fill_batch call —
what splitting a chip-filling list inside the call could buy.  usage: split_probe.py [NGAPS] [K ...]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B
from gap2seq_amd import lib as P

ngaps = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
ks = [int(x) for x in sys.argv[2:]] or [2, 3, 4, 5]
reads = P.G2S.synth_genome(3000000, 3, B.GENOME_SEED)
seqs = [ln for ln in reads.splitlines() if not ln.startswith(">")]
gaps = B.parse_gaps(P.G2S.synth_gaps(reads, 31, 10, ngaps, 200, 1000, B.GAP_SEED), 10)
graph = P.Graph.from_seqs(seqs, 31, 1)
s = P.Session(graph, 0, d_err=500, randseed=1)
lib = P.load_library()
arr, keep = P._gap_array([P.Gap(g["left"], g["right"], g["gap_len"], g["lmf"], g["rmf"]) for g in gaps])
n = len(gaps)
nbytes = lib.g2s_team_arena_bytes(s.h, arr, n)
arena = P.HostBuffer(nbytes)
rbuf = P.HostBuffer(C.sizeof(P.g2s_result) * n)
res = rbuf.array(P.g2s_result, n)
ap = C.cast(arena.p, C.c_void_p)
gsz = C.sizeof(P.g2s_gap)
rsz = C.sizeof(P.g2s_result)


def one():
    s.srand(1)
    t0 = time.perf_counter()
    P._check(lib.g2s_fill_batch(s.h, arr, n, res, C.cast(arena.p, C.c_char_p), nbytes))
    return time.perf_counter() - t0


def key():
    raw = arena.raw
    return [(r.count, r.left_fuz, r.right_fuz, r.fill_len, r.draws) for r in (res[i] for i in range(n))], raw


def split(K):
    s.srand(1)
    bounds = [n * q // K for q in range(K + 1)]
    subs = []
    off = 0
    for q in range(K):
        lo, hi = bounds[q], bounds[q + 1]
        sub = C.cast(C.addressof(arr) + lo * gsz, C.POINTER(P.g2s_gap))
        nb = lib.g2s_team_arena_bytes(s.h, sub, hi - lo)
        subs.append((sub, hi - lo, C.cast(C.addressof(res) + lo * rsz, C.POINTER(P.g2s_result)), C.c_void_p(arena.p + off), nb))
        off += nb
    t0 = time.perf_counter()
    inflight = 0
    for q in range(K):
        sub, m, r, a, nb = subs[q]
        if inflight == P.G2S_MAX_IN_FLIGHT:
            P._check(lib.g2s_fill_end(s.h)); inflight -= 1
        P._check(lib.g2s_fill_begin(s.h, sub, m, r, a, nb)); inflight += 1
    while inflight:
        P._check(lib.g2s_fill_end(s.h)); inflight -= 1
    return time.perf_counter() - t0


for _ in range(5):
    one()
t1 = min(one() for _ in range(20))
k1, raw1 = key()
print("one call: %.3f ms (best of 20), %.2f M gaps/s" % (t1 * 1e3, n / t1 / 1e6))
for K in ks:
    for _ in range(5):
        split(K)
    ts = sorted(split(K) for _ in range(20))
    kk, _ = key()
    same = [x[:3] + x[4:] for x in kk] == [x[:3] + x[4:] for x in k1]
    print("K=%d sub-lists, %d in flight: best %.3f ms median %.3f ms, %.2f M gaps/s; counts/fuz/draws identical: %s"
          % (K, min(K, P.G2S_MAX_IN_FLIGHT), ts[0] * 1e3, ts[len(ts) // 2] * 1e3, n / ts[0] / 1e6, same))
