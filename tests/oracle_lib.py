"""tests/oracle_lib.py — ctypes binding of the CPU oracle (oracle/_build/libg2s_oracle.so).
TEST INFRASTRUCTURE ONLY: imported by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke(); never by the product package."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "_build", "libg2s_oracle.so")


class orc_info(C.Structure):
    _fields_ = [("sub", C.c_uint64 * 6), ("ctr", C.c_uint64 * 6), ("phaseC_count", C.c_int32),
                ("n_lengths", C.c_int32), ("lengths", C.c_int32 * 2), ("reached_fuz", C.c_int32),
                ("draws", C.c_int32), ("q7", C.c_int32), ("backtrace_failed", C.c_int32),
                ("mem_exceeded", C.c_int32), ("final_d", C.c_int32), ("pad", C.c_int32),
                ("max_border_a", C.c_int32), ("max_border_b", C.c_int32)]


class orc_params(C.Structure):
    _fields_ = [("k", C.c_int32), ("solid", C.c_int32), ("d_err", C.c_int32), ("max_fuz", C.c_int32),
                ("max_mem_gb", C.c_double), ("skip_confident", C.c_int32), ("unique_paths", C.c_int32),
                ("all_paths", C.c_int32), ("randseed", C.c_int32), ("nb_cores", C.c_int32)]


class orc_summary(C.Structure):
    _fields_ = [("gaps", C.c_int32), ("filled", C.c_int32), ("q7_gaps", C.c_int32), ("pad", C.c_int32),
                ("ctr", C.c_uint64 * 6), ("fill_seconds", C.c_double)]


_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(ORACLE_SO):
            build()
        L = C.CDLL(ORACLE_SO)
        VP = C.c_void_p
        L.orc_graph_from_seqs.restype = VP
        L.orc_graph_from_seqs.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int]
        L.orc_graph_from_files.restype = VP
        L.orc_graph_from_files.argtypes = [C.c_char_p, C.c_int, C.c_int]
        L.orc_graph_free.argtypes = [VP]
        L.orc_graph_num_kmers.restype = C.c_uint64
        L.orc_graph_num_kmers.argtypes = [VP]
        L.orc_rng_new.restype = VP
        L.orc_rng_new.argtypes = [C.c_uint]
        L.orc_rng_free.argtypes = [VP]
        L.orc_rng_next.argtypes = [VP]
        L.orc_rng_skip.argtypes = [VP, C.c_ulonglong]
        fill_args = [VP, VP, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_longlong, C.c_int,
                     C.c_int, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(orc_info)]
        L.orc_fill_gap.argtypes = fill_args
        L.orc_fill_gap_dump.argtypes = fill_args + [C.POINTER(VP)]
        L.orc_free_str.argtypes = [VP]
        L.orc_execute_scaffolds.argtypes = [VP, C.POINTER(orc_params), C.c_char_p, C.c_char_p, C.c_char_p,
                                            C.POINTER(VP), C.POINTER(VP), C.POINTER(orc_summary)]
        L.orc_execute_single.argtypes = [VP, C.POINTER(orc_params), C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p,
                                         C.c_int, C.POINTER(VP), C.POINTER(VP)]
        L.orc_time_fill_batch.restype = C.c_double
        L.orc_time_fill_batch.argtypes = [VP, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_int),
                                          C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_uint64)]
        _LIB = L
    return _LIB


def time_fill_batch(graph, gaps, gap_err, nthreads=1, skip_confident=False, all_paths=True):
    """CPU baseline timing: (seconds, filled, counters[6]) for fill_gap over `gaps`
    (dicts with left/right/gap_len/lmf/rmf)."""
    L = lib()
    n = len(gaps)
    lefts = (C.c_char_p * n)(*[g["left"].encode() for g in gaps])
    rights = (C.c_char_p * n)(*[g["right"].encode() for g in gaps])
    gl = (C.c_int * n)(*[g["gap_len"] for g in gaps])
    lm = (C.c_int * n)(*[g["lmf"] for g in gaps])
    rm = (C.c_int * n)(*[g["rmf"] for g in gaps])
    filled = C.c_int(0)
    ctr = (C.c_uint64 * 6)()
    secs = L.orc_time_fill_batch(graph.h, lefts, rights, gl, lm, rm, n, gap_err, int(skip_confident), int(all_paths),
                                 nthreads, C.byref(filled), ctr)
    return secs, filled.value, [int(x) for x in ctr]


class OracleGraph:
    def __init__(self, seqs, k, solid):
        L = lib()
        enc = [s.encode("ascii") for s in seqs]
        arr = (C.c_char_p * len(enc))(*enc)
        self.h = L.orc_graph_from_seqs(arr, len(enc), k, solid)
        self.k = k
        assert self.h

    @property
    def num_kmers(self):
        return lib().orc_graph_num_kmers(self.h)

    def free(self):
        if self.h:
            lib().orc_graph_free(self.h)
            self.h = None


class OracleRng:
    """srand(seed), optionally advanced by `skip` draws."""

    def __init__(self, seed, skip=0):
        self.h = lib().orc_rng_new(seed)
        if skip:
            lib().orc_rng_skip(self.h, skip)

    def next(self):
        return lib().orc_rng_next(self.h)

    def skip(self, n):
        lib().orc_rng_skip(self.h, n)

    def free(self):
        if self.h:
            lib().orc_rng_free(self.h)
            self.h = None


class OracleFill:
    pass


def fill_gap(graph, rng, left, right, gap_len, gap_err, lmf, rmf, skip_confident=False, all_paths=True,
             max_mem=1 << 50, dump=False):
    """One oracle fill_gap call -> OracleFill(count, left_fuz, right_fuz, fill (from lmf-left_fuz), info, states)."""
    L = lib()
    buf = C.create_string_buffer(gap_len + graph.k + gap_err + lmf + rmf + 3)
    lf, rf = C.c_int(0), C.c_int(0)
    info = orc_info()
    o = OracleFill()
    o.states = None
    if dump:
        st = C.c_void_p()
        o.count = L.orc_fill_gap_dump(graph.h, rng.h, left.encode(), right.encode(), gap_len, gap_err, lmf, rmf,
                                      max_mem, int(skip_confident), int(all_paths), buf, C.byref(lf), C.byref(rf),
                                      C.byref(info), C.byref(st))
        text = C.string_at(st).decode()
        L.orc_free_str(st)
        o.states = [(p[0], int(p[1]), int(p[2])) for p in (ln.split() for ln in text.splitlines())]
    else:
        o.count = L.orc_fill_gap(graph.h, rng.h, left.encode(), right.encode(), gap_len, gap_err, lmf, rmf, max_mem,
                                 int(skip_confident), int(all_paths), buf, C.byref(lf), C.byref(rf), C.byref(info))
    o.left_fuz, o.right_fuz = lf.value, rf.value
    o.phase_d = info.phaseC_count > 0 and info.n_lengths > 0
    o.fill = buf.raw[lmf - lf.value:].split(b"\0")[0].decode() if o.phase_d else ""
    o.info = info
    o.substats = list(info.sub)
    o.lengths = [info.lengths[i] for i in range(info.n_lengths)]
    return o


def execute_scaffolds(graph, scaffolds_text, k, solid=2, d_err=500, max_fuz=10, randseed=1, skip_confident=False,
                      unique_paths=False, all_paths=True, nb_cores=1, max_mem_gb=20.0, reads_label="reads.fa",
                      filled_label="filled.fa"):
    L = lib()
    p = orc_params(k, solid, d_err, max_fuz, max_mem_gb, int(skip_confident), int(unique_paths), int(all_paths),
                   randseed, nb_cores)
    fa, lg = C.c_void_p(), C.c_void_p()
    sm = orc_summary()
    L.orc_execute_scaffolds(graph.h, C.byref(p), reads_label.encode(), filled_label.encode(),
                            scaffolds_text.encode("ascii"), C.byref(fa), C.byref(lg), C.byref(sm))
    fasta, log = C.string_at(fa).decode(), C.string_at(lg).decode()
    L.orc_free_str(fa)
    L.orc_free_str(lg)
    return fasta, log, sm


def execute_single(graph, left, right, length, k, solid=2, d_err=500, max_fuz=10, randseed=1, skip_confident=False,
                   unique_paths=False, all_paths=True, max_mem_gb=20.0, reads_label="reads.fa",
                   filled_label="filled.fa"):
    L = lib()
    p = orc_params(k, solid, d_err, max_fuz, max_mem_gb, int(skip_confident), int(unique_paths), int(all_paths),
                   randseed, 1)
    fa, lg = C.c_void_p(), C.c_void_p()
    L.orc_execute_single(graph.h, C.byref(p), reads_label.encode(), filled_label.encode(), left.encode(),
                         right.encode(), length, C.byref(fa), C.byref(lg))
    fasta, log = C.string_at(fa).decode(), C.string_at(lg).decode()
    L.orc_free_str(fa)
    L.orc_free_str(lg)
    return fasta, log
