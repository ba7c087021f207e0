"""ctypes binding of include/g2s.h (the drop-in boundary for
/root/reference/src/Gap2Seq.cpp:858 ``Gap2Seq::fill_gap`` and its caller
``Gap2Seq::execute`` :161-438).  Names follow the C ABI one to one.

No compute lives here.  If ``libg2s_hip.so`` is missing the import of the library
raises; if no gfx950 device is usable every fill call raises G2SError
(G2S_ERR_NO_DEVICE) — there is no CPU path to fall back to.
"""
import ctypes as C
import os

G2S_OK = 0
G2S_ERR_IO = -2
G2S_ERR_NO_DEVICE = -3
G2S_ERR_STATE = -6
G2S_INVALID_NODE = 0xFFFFFFFF
G2S_MAX_PATHS = 2147483647 // 2 - 1
G2S_MAX_IN_FLIGHT = 3  # include/g2s.h: lists begun and not ended on one session
G2S_GROUP_PER_SESSION = (1 << 64) - 1  # include/g2s.h: g2s_session_set_team cuts every long list into one group per session
G2S_GAP_SKIPPED = 0x1
G2S_GAP_Q7 = 0x2
G2S_GAP_MEM_EXCEEDED = 0x4
G2S_GAP_BACKTRACE_FAIL = 0x8
G2S_GAP_BAD_FLANK = 0x10
G2S_GAP_PHASE_D = 0x20


class G2SError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("g2s error %d: %s" % (code, msg))
        self.code = code


class g2s_params(C.Structure):
    _fields_ = [("d_err", C.c_int32), ("skip_confident", C.c_int32), ("all_paths", C.c_int32),
                ("unique_paths", C.c_int32), ("max_mem", C.c_int64), ("randseed", C.c_uint32),
                ("host_threads", C.c_int32)]


class g2s_gap(C.Structure):
    _fields_ = [("left", C.c_char_p), ("right", C.c_char_p), ("left_len", C.c_int32), ("right_len", C.c_int32),
                ("gap_len", C.c_int32), ("lmf", C.c_int32), ("rmf", C.c_int32),
                ("skip_if_prev_right_fuz_gt", C.c_int32)]


class g2s_result(C.Structure):
    _fields_ = [("count", C.c_int32), ("left_fuz", C.c_int32), ("right_fuz", C.c_int32), ("flags", C.c_uint32),
                ("fill_off", C.c_uint64), ("fill_len", C.c_int32), ("draws", C.c_int32),
                ("vertices", C.c_uint64), ("edges", C.c_uint64), ("nontrivial_components", C.c_uint64),
                ("size_nontrivial_components", C.c_uint64), ("vertices_final", C.c_uint64),
                ("edges_final", C.c_uint64), ("phaseC_count", C.c_int32), ("n_lengths", C.c_int32),
                ("lengths", C.c_int32 * 2), ("backtrace_depth", C.c_int32), ("backtrace_final_d", C.c_int32),
                ("reserved", C.c_int32 * 2)]


class g2s_timing(C.Structure):
    _fields_ = [("ms_right_bfs", C.c_double), ("ms_left_dp", C.c_double), ("ms_extract", C.c_double),
                ("ms_d2h", C.c_double), ("ms_host_post", C.c_double), ("ms_total", C.c_double),
                ("xA", C.c_uint64), ("sA", C.c_uint64), ("xB", C.c_uint64), ("sB", C.c_uint64),
                ("xD", C.c_uint64), ("sD", C.c_uint64), ("flank_bytes", C.c_uint64), ("fill_bytes", C.c_uint64),
                ("launches_left_dp", C.c_uint32), ("retried_gaps", C.c_uint32),
                ("ms_fill_lds", C.c_double), ("ms_extract_lds", C.c_double), ("x_fill_lds", C.c_uint64),
                ("s_fill_lds", C.c_uint64), ("lds_tier_gaps", C.c_uint32), ("lds_launches", C.c_uint32),
                ("log_pool_gaps", C.c_uint32), ("rs_pool_gaps", C.c_uint32), ("ms_prepare", C.c_double),
                ("ms_fill_seg", C.c_double), ("seg_tier_gaps", C.c_uint32), ("seg_launches", C.c_uint32),
                ("seg_segments", C.c_uint64), ("ms_fill_segx", C.c_double), ("segx_tier_gaps", C.c_uint32),
                ("segx_launches", C.c_uint32), ("watchdog_gaps", C.c_uint32), ("seg2_launches", C.c_uint32),
                ("ms_d3", C.c_double), ("resident_launches", C.c_uint32), ("resident_fallbacks", C.c_uint32),
                ("draw_dependent_gaps", C.c_uint64), ("d3_table_entries", C.c_uint64),
                ("host_finished_gaps", C.c_uint32), ("team_groups", C.c_uint32), ("team_sessions", C.c_uint32),
                ("team_groups_by_session", C.c_uint32 * 16), ("seg_timed_launches", C.c_uint32), ("team_d3_sharded", C.c_uint32), ("traced_in_fill_gaps", C.c_uint32),
                ("team_ms_fill", C.c_double * 16), ("team_ms_d3", C.c_double * 16), ("team_ms_wall", C.c_double * 16),
                ("host_us", C.c_double * 8), ("guessed_in_fill_gaps", C.c_uint32), ("guessed_groups", C.c_uint32),
                ("guessed_groups_resent", C.c_uint32), ("reserved1", C.c_uint32)]


class g2s_run_opts(C.Structure):
    _fields_ = [("k", C.c_int32), ("solid", C.c_int32), ("max_fuz", C.c_int32), ("nb_cores", C.c_int32),
                ("max_mem_gb", C.c_double)]


class g2s_filter_opts(C.Structure):
    _fields_ = [("mean_insert", C.c_int32), ("std_dev", C.c_int32), ("breakpoint", C.c_int32), ("gap_length", C.c_int32),
                ("flank_length", C.c_int32), ("unmapped_only", C.c_int32), ("threads", C.c_int32),
                ("scaffold", C.c_char_p)]


# every symbol include/g2s.h declares: name -> (restype, argtypes)
_VP = C.c_void_p
TEXT_FN = C.CFUNCTYPE(None, C.POINTER(C.c_char), C.c_size_t, C.c_void_p)  # g2s_text_fn
_SIGS = {
    "g2s_abi_version": (C.c_int, []),
    "g2s_fill_begin": (C.c_int, [_VP, C.POINTER(g2s_gap), C.c_size_t, C.POINTER(g2s_result), C.c_void_p, C.c_size_t]),
    "g2s_fill_end": (C.c_int, [_VP]),
    "g2s_fill_in_flight": (C.c_int, [_VP]),
    "g2s_share_begin": (C.c_int, [_VP, C.POINTER(g2s_gap), C.c_size_t, C.POINTER(g2s_result), C.c_void_p, C.c_size_t,
                                  C.POINTER(C.c_uint64)]),
    "g2s_share_tables": (C.c_int, [_VP, C.c_uint64, C.c_uint64, C.POINTER(C.POINTER(C.c_uint32))]),
    "g2s_share_trace": (C.c_int, [_VP, C.c_uint32]),
    "g2s_share_end": (C.c_int, [_VP, C.c_uint64]),
    "g2s_backtrace_text": (C.c_size_t, [C.POINTER(g2s_gap), C.POINTER(g2s_result), C.c_int, C.c_char_p, C.c_size_t]),
    "g2s_last_error": (C.c_char_p, []),
    "g2s_graph_build_files": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(_VP)]),
    "g2s_graph_build_seqs": (C.c_int, [C.POINTER(C.c_char_p), C.POINTER(C.c_uint64), C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.POINTER(_VP)]),
    "g2s_graph_save": (C.c_int, [_VP, C.c_char_p]),
    "g2s_graph_load": (C.c_int, [C.c_char_p, C.POINTER(_VP)]),
    "g2s_graph_free": (None, [_VP]),
    "g2s_graph_k": (C.c_int, [_VP]),
    "g2s_graph_solid": (C.c_int, [_VP]),
    "g2s_graph_num_kmers": (C.c_uint64, [_VP]),
    "g2s_graph_num_unitigs": (C.c_uint64, [_VP]),
    "g2s_graph_node": (C.c_uint32, [_VP, C.c_char_p]),
    "g2s_graph_successors": (C.c_int, [_VP, C.c_uint32, C.POINTER(C.c_uint32)]),
    "g2s_graph_predecessors": (C.c_int, [_VP, C.c_uint32, C.POINTER(C.c_uint32)]),
    "g2s_graph_node_string": (C.c_int, [_VP, C.c_uint32, C.c_char_p]),
    "g2s_graph_upload": (C.c_int, [_VP, C.c_int]),
    "g2s_graph_device_bytes": (C.c_uint64, [_VP, C.c_int]),
    "g2s_session_create": (C.c_int, [_VP, C.c_int, C.POINTER(g2s_params), C.POINTER(_VP)]),
    "g2s_session_destroy": (None, [_VP]),
    "g2s_session_srand": (C.c_int, [_VP, C.c_uint32]),
    "g2s_session_skip_draws": (C.c_int, [_VP, C.c_uint64]),
    "g2s_session_graph": (_VP, [_VP]),
    "g2s_session_get_params": (C.c_int, [_VP, C.POINTER(g2s_params)]),
    "g2s_batch_prepare": (C.c_int, [_VP, C.POINTER(g2s_gap), C.c_size_t, C.POINTER(_VP)]),
    "g2s_batch_run": (C.c_int, [_VP, C.POINTER(g2s_result), C.c_char_p, C.c_size_t]),
    "g2s_batch_arena_bytes": (C.c_size_t, [_VP]),
    "g2s_batch_timing": (C.c_int, [_VP, C.POINTER(g2s_timing)]),
    "g2s_batch_free": (None, [_VP]),
    "g2s_session_last_timing": (C.c_int, [_VP, C.POINTER(g2s_timing)]),
    "g2s_fill_batch": (C.c_int, [_VP, C.POINTER(g2s_gap), C.c_size_t, C.POINTER(g2s_result), C.c_char_p,
                                 C.c_size_t]),
    "g2s_team_fill": (C.c_int, [C.POINTER(_VP), C.c_int, C.POINTER(g2s_gap), C.c_size_t, C.c_size_t,
                                C.POINTER(g2s_result), C.c_char_p, C.c_size_t, C.POINTER(g2s_timing)]),
    "g2s_team_arena_bytes": (C.c_size_t, [_VP, C.POINTER(g2s_gap), C.c_size_t]),
    "g2s_session_set_team": (C.c_int, [_VP, C.POINTER(_VP), C.c_int, C.c_size_t]),
    "g2s_execute_scaffolds_stream": (C.c_int, [_VP, C.POINTER(g2s_run_opts), C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t,
                                             TEXT_FN, TEXT_FN, _VP, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "g2s_execute_scaffolds": (C.c_int, [_VP, C.POINTER(g2s_run_opts), C.c_char_p, C.c_char_p, C.c_char_p,
                                        C.POINTER(_VP), C.POINTER(_VP), C.POINTER(C.c_int32),
                                        C.POINTER(C.c_int32)]),
    "g2s_execute_single": (C.c_int, [_VP, C.POINTER(g2s_run_opts), C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p,
                                     C.c_int32, C.POINTER(_VP), C.POINTER(_VP)]),
    "g2s_free": (None, [_VP]),
    "g2s_host_alloc": (_VP, [C.c_size_t]),
    "g2s_host_free": (None, [_VP]),
    "g2s_cut_scaffolds": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p,
                                    C.c_char_p, C.POINTER(_VP), C.POINTER(_VP), C.POINTER(_VP), C.POINTER(_VP)]),
    "g2s_merge_scaffolds": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(_VP),
                                      C.POINTER(_VP)]),
    "g2s_filter_reads": (C.c_int, [C.c_char_p, C.POINTER(g2s_filter_opts), C.POINTER(_VP), C.POINTER(_VP),
                                   C.POINTER(_VP), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "g2s_filter_reads_mem": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(g2s_filter_opts), C.POINTER(_VP),
                                       C.POINTER(_VP), C.POINTER(_VP), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "g2s_filter_last_error": (C.c_char_p, []),
    "g2s_device_count": (C.c_int, []),
    "g2s_synth_genome": (C.c_int, [C.c_uint64, C.c_uint32, C.c_uint64, C.POINTER(_VP)]),
    "g2s_synth_gaps": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64,
                                 C.POINTER(_VP)]),
    "g2s_test_rand_skip": (C.c_int, [C.c_uint32, C.c_uint64, C.c_uint32, C.POINTER(C.c_int32)]),
    "g2s_test_rand_stream": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_int32)]),
    "g2s_test_device_rand": (C.c_int, [C.c_int, C.c_uint32, C.c_uint64, C.c_uint32, C.POINTER(C.c_int32)]),
    "g2s_test_post_closure": (C.c_int, [_VP, C.POINTER(g2s_params), C.POINTER(g2s_gap), C.c_uint32,
                                        C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_uint64), C.c_int32, C.c_int32,
                                        C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_uint32, C.c_uint64,
                                        C.POINTER(g2s_result), C.c_char_p]),
    "g2s_test_graph_tables": (C.c_int, [_VP, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    "g2s_test_post_segments": (C.c_int, [_VP, C.POINTER(g2s_params), C.POINTER(g2s_gap), C.c_uint32,
                                         C.POINTER(C.c_uint32), C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_int32,
                                         C.c_int32, C.c_uint32, C.c_uint64, C.POINTER(g2s_result), C.c_char_p,
                                         C.POINTER(C.c_int32)]),
    "g2s_test_seg_expand": (C.c_int, [_VP, C.POINTER(g2s_params), C.POINTER(g2s_gap), C.c_uint32, C.POINTER(C.c_uint32),
                                      C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.c_uint32, C.POINTER(C.c_uint32),
                                      C.c_uint32, C.POINTER(C.c_uint64)]),
    "g2s_test_worker_pool": (C.c_int, [C.c_int32, C.c_int32, C.c_int32]),
    "g2s_test_group_queue": (C.c_int, [C.c_int32, C.c_uint64, C.c_uint64, C.POINTER(C.c_int32)]),
    "g2s_test_group_queue_slow": (C.c_int, [C.c_int32, C.c_uint64, C.c_uint64, C.c_int32, C.c_uint32, C.POINTER(C.c_int32)]),
    "g2s_graph_validate": (C.c_int64, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "g2s_test_post_gap": (C.c_int, [_VP, C.POINTER(g2s_params), C.POINTER(g2s_gap), C.c_int32,
                                    C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.c_int32,
                                    C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_uint32, C.c_uint32,
                                    C.POINTER(g2s_result), C.c_char_p]),
}


def library_path():
    # (G2S_LIBRARY: an instrumented build of the same sources, tools/ only — the product library stays where it is)
    return os.environ.get("G2S_LIBRARY") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libg2s_hip.so")


_LIB = None


def load_library():
    """Load the in-tree HIP extension; raises OSError when it has not been built."""
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            raise OSError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc, gfx950). There is no CPU fallback." % path)
        lib = C.CDLL(path)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)  # AttributeError = ABI symbol missing
            fn.restype = res
            fn.argtypes = args
        _LIB = lib
    return _LIB


def _check(rc):
    if rc != G2S_OK:
        raise G2SError(rc, (load_library().g2s_last_error() or b"").decode("utf-8", "replace"))


def _take_text(ptr):
    lib = load_library()
    if not ptr:
        return ""
    s = C.string_at(ptr).decode("ascii")
    lib.g2s_free(ptr)
    return s


class G2S:
    """Namespace for the free functions of the ABI."""

    @staticmethod
    def device_count():
        return load_library().g2s_device_count()

    @staticmethod
    def synth_genome(length, variant, seed):
        out = _VP()
        _check(load_library().g2s_synth_genome(length, variant, seed, C.byref(out)))
        return _take_text(out)

    @staticmethod
    def synth_gaps(reads_fasta, k, fuz, ngaps, min_len, max_len, seed):
        out = _VP()
        _check(load_library().g2s_synth_gaps(reads_fasta.encode("ascii"), k, fuz, ngaps, min_len, max_len, seed,
                                             C.byref(out)))
        return _take_text(out)


def cut_scaffolds(scaffolds_text, k=31, fuz=10, mask=False, no_split=False, labels=("scaffolds.fa", "contigs.fa", "gaps.fa", "gaps.bed")):
    """g2s_cut_scaffolds (GapCutter): returns (contigs_fasta, gaps_fasta, bed, log)."""
    outs = [_VP(), _VP(), _VP(), _VP()]
    _check(load_library().g2s_cut_scaffolds(scaffolds_text.encode("ascii"), k, fuz, int(mask), int(no_split),
                                            *[x.encode() for x in labels], *[C.byref(o) for o in outs]))
    return tuple(_take_text(o) for o in outs)


def merge_scaffolds(contigs_text, gaps_text, labels=("merged.fa", "contigs.fa", "filled.fa")):
    """g2s_merge_scaffolds (GapMerger): returns (scaffolds_fasta, log)."""
    out, log = _VP(), _VP()
    _check(load_library().g2s_merge_scaffolds(contigs_text.encode("ascii"), gaps_text.encode("ascii"),
                                              *[x.encode() for x in labels], C.byref(out), C.byref(log)))
    return _take_text(out), _take_text(log)


def filter_reads(bam, mean, std_dev, scaffold, breakpoint, gap_length=-1, flank_length=-1, unmapped_only=False,
                 threads=0):
    """g2s_filter_reads / g2s_filter_reads_mem (ReadFilter): `bam` is a path (str) or the file's bytes.
    Returns (fasta, stdout_text, stderr_text, extracted, total)."""
    lib = load_library()
    o = g2s_filter_opts(mean, std_dev, breakpoint, gap_length, flank_length, int(unmapped_only), threads,
                        scaffold.encode())
    outs = [_VP(), _VP(), _VP()]
    ext, tot = C.c_int64(0), C.c_int64(0)
    tail = [C.byref(o)] + [C.byref(x) for x in outs] + [C.byref(ext), C.byref(tot)]
    if isinstance(bam, (bytes, bytearray)):
        rc = lib.g2s_filter_reads_mem(bytes(bam), len(bam), *tail)
    else:
        rc = lib.g2s_filter_reads(str(bam).encode(), *tail)
    if rc != G2S_OK:
        raise G2SError(rc, (lib.g2s_filter_last_error() or b"").decode("utf-8", "replace"))
    texts = []
    for x in outs:  # (read names are bytes of the BAM file: not necessarily ASCII)
        texts.append(C.string_at(x).decode("latin-1") if x else "")
        lib.g2s_free(x)
    return tuple(texts) + (ext.value, tot.value)


class Graph:
    """g2s_graph: replaces gatb Graph::create / Graph::load (Gap2Seq.cpp:193-219)."""

    def __init__(self, handle):
        self.h = handle

    @classmethod
    def from_seqs(cls, seqs, k, solid, nthreads=0):
        lib = load_library()
        enc = [s.encode("ascii") if isinstance(s, str) else s for s in seqs]
        arr = (C.c_char_p * len(enc))(*enc)
        lens = (C.c_uint64 * len(enc))(*[len(e) for e in enc])
        h = _VP()
        _check(lib.g2s_graph_build_seqs(arr, lens, len(enc), k, solid, nthreads, C.byref(h)))
        return cls(h)

    @classmethod
    def from_files(cls, reads_csv, k, solid, nthreads=0):
        h = _VP()
        _check(load_library().g2s_graph_build_files(reads_csv.encode(), k, solid, nthreads, C.byref(h)))
        return cls(h)

    @classmethod
    def load(cls, path):
        h = _VP()
        _check(load_library().g2s_graph_load(path.encode(), C.byref(h)))
        return cls(h)

    def save(self, path):
        _check(load_library().g2s_graph_save(self.h, path.encode()))

    @property
    def k(self):
        return load_library().g2s_graph_k(self.h)

    @property
    def num_kmers(self):
        return load_library().g2s_graph_num_kmers(self.h)

    @property
    def num_unitigs(self):
        return load_library().g2s_graph_num_unitigs(self.h)

    def validate(self):
        """(violations, text): bitmap / successor table / last-base table consistency (test hook)."""
        buf = C.create_string_buffer(2048)
        n = load_library().g2s_graph_validate(self.h, buf, 2048)
        return n, buf.value.decode()

    def node(self, kmer):
        return load_library().g2s_graph_node(self.h, kmer.encode("ascii"))

    def node_string(self, v):
        buf = C.create_string_buffer(self.k + 1)
        _check(load_library().g2s_graph_node_string(self.h, v, buf))
        return buf.value.decode()

    def successors(self, v):
        out = (C.c_uint32 * 4)()
        n = load_library().g2s_graph_successors(self.h, v, out)
        return [out[i] for i in range(n)]

    def predecessors(self, v):
        out = (C.c_uint32 * 4)()
        n = load_library().g2s_graph_predecessors(self.h, v, out)
        return [out[i] for i in range(n)]

    def upload(self, device=0):
        _check(load_library().g2s_graph_upload(self.h, device))

    def device_bytes(self, device=0):
        return load_library().g2s_graph_device_bytes(self.h, device)

    def free(self):
        if self.h:
            load_library().g2s_graph_free(self.h)
            self.h = None


class Gap:
    """Arguments of one fill_gap call (Gap2Seq.cpp:380-383)."""

    def __init__(self, left, right, gap_len, lmf, rmf, skip_if_prev_right_fuz_gt=-1):
        self.left, self.right, self.gap_len, self.lmf, self.rmf = left, right, gap_len, lmf, rmf
        self.skip_dep = skip_if_prev_right_fuz_gt


def make_params(d_err=500, skip_confident=False, all_paths=True, unique_paths=False, max_mem=20 << 30, randseed=1,
                host_threads=0):
    return g2s_params(d_err, int(skip_confident), int(all_paths), int(unique_paths), max_mem, randseed, host_threads)


def _gap_array(gaps):
    keep = []
    arr = (g2s_gap * max(1, len(gaps)))()
    for i, g in enumerate(gaps):
        l, r = g.left.encode("ascii"), g.right.encode("ascii")
        keep.append((l, r))
        arr[i] = g2s_gap(l, r, len(l), len(r), g.gap_len, g.lmf, g.rmf, g.skip_dep)
    return arr, keep


class FillResult:
    """g2s_result with the fill text resolved."""

    def __init__(self, r, arena):
        for name, _ in g2s_result._fields_:
            if name not in ("lengths", "reserved"):
                setattr(self, name, getattr(r, name))
        self.lengths = [r.lengths[i] for i in range(r.n_lengths)]
        # (the reference's "Unable to backtrace!" line minus the k-mer, which g2s_backtrace_text reads from the gap)
        self.backtrace_msg = ("Unable to backtrace! %d %d" % (r.backtrace_depth, r.backtrace_final_d)) if r.flags & G2S_GAP_BACKTRACE_FAIL else ""
        self.fill = arena[r.fill_off:r.fill_off + r.fill_len].decode("ascii") if r.fill_len > 0 else ""
        self.substats = [r.vertices, r.edges, r.nontrivial_components, r.size_nontrivial_components,
                         r.vertices_final, r.edges_final]


class HostBuffer:
    """g2s_host_alloc: page-locked memory the kernels write directly (results, fill arena)."""

    def __init__(self, nbytes):
        self.nbytes = max(16, int(nbytes))
        self.p = load_library().g2s_host_alloc(self.nbytes)
        if not self.p:
            raise MemoryError("g2s_host_alloc(%d) failed" % self.nbytes)

    def array(self, ctype, count):
        return (ctype * count).from_address(self.p)

    @property
    def raw(self):
        return C.string_at(self.p, self.nbytes)

    def free(self):
        if self.p:
            load_library().g2s_host_free(self.p)
            self.p = None


class Session:
    """g2s_session: one GPU, one rand() stream."""

    def __init__(self, graph, device=0, **kw):
        self.graph = graph
        self.params = make_params(**kw)
        h = _VP()
        _check(load_library().g2s_session_create(graph.h, device, C.byref(self.params), C.byref(h)))
        self.h = h

    def last_timing(self):
        """g2s_session_last_timing: the g2s_timing of the last list this session finished."""
        t = g2s_timing()
        _check(load_library().g2s_session_last_timing(self.h, C.byref(t)))
        return t

    def srand(self, seed, skip=0):
        _check(load_library().g2s_session_srand(self.h, seed))
        if skip:
            _check(load_library().g2s_session_skip_draws(self.h, skip))

    def fill_batch(self, gaps, want_timing=False, pinned=False):
        """prepare + run; returns list of FillResult (and g2s_timing).  pinned: results and arena in
        g2s_host_alloc memory (the kernels then write them directly when the list is finished on the device)."""
        lib = load_library()
        arr, keep = _gap_array(gaps)
        b = _VP()
        _check(lib.g2s_batch_prepare(self.h, arr, len(gaps), C.byref(b)))
        bufs = []
        try:
            nbytes = lib.g2s_batch_arena_bytes(b)
            if pinned:
                bufs = [HostBuffer(max(1, nbytes)), HostBuffer(C.sizeof(g2s_result) * max(1, len(gaps)))]
                arena = bufs[0]
                res = bufs[1].array(g2s_result, max(1, len(gaps)))
                _check(lib.g2s_batch_run(b, res, C.cast(arena.p, C.c_char_p), nbytes))
            else:
                arena = C.create_string_buffer(max(1, nbytes))
                res = (g2s_result * max(1, len(gaps)))()
                _check(lib.g2s_batch_run(b, res, arena, nbytes))
            t = g2s_timing()
            _check(lib.g2s_batch_timing(b, C.byref(t)))
            raw = arena.raw
            out = [FillResult(res[i], raw) for i in range(len(gaps))]
        finally:
            lib.g2s_batch_free(b)
            for hb in bufs:
                hb.free()
        return (out, t) if want_timing else out

    def fill_lists_overlapped(self, lists, pinned=True, depth=G2S_MAX_IN_FLIGHT):
        """g2s_fill_begin / g2s_fill_end over consecutive lists, `depth` in flight: list i+depth-1 is begun before
        list i is ended.  Returns the lists' results in order (and the timing of the last list ended)."""
        lib = load_library()
        ctx = []
        for gaps in lists:
            arr, keep = _gap_array(gaps)
            nbytes = lib.g2s_team_arena_bytes(self.h, arr, len(gaps))
            if pinned:
                arena = HostBuffer(max(1, nbytes))
                rbuf = HostBuffer(C.sizeof(g2s_result) * max(1, len(gaps)))
                res = rbuf.array(g2s_result, max(1, len(gaps)))
                ctx.append(dict(arr=arr, keep=keep, n=len(gaps), nbytes=nbytes, arena=arena, rbuf=rbuf, res=res, ap=C.cast(arena.p, C.c_void_p)))
            else:
                arena = C.create_string_buffer(max(1, nbytes))
                res = (g2s_result * max(1, len(gaps)))()
                ctx.append(dict(arr=arr, keep=keep, n=len(gaps), nbytes=nbytes, arena=arena, rbuf=None, res=res, ap=C.cast(arena, C.c_void_p)))
        out = []
        t = g2s_timing()
        try:
            for i, c in enumerate(ctx):
                _check(lib.g2s_fill_begin(self.h, c["arr"], c["n"], c["res"], c["ap"], c["nbytes"]))
                if i >= depth - 1:
                    _check(lib.g2s_fill_end(self.h))
            while lib.g2s_fill_in_flight(self.h) > 0:
                _check(lib.g2s_fill_end(self.h))
            _check(lib.g2s_session_last_timing(self.h, C.byref(t)))
            for c in ctx:
                raw = c["arena"].raw
                out.append([FillResult(c["res"][i], raw) for i in range(c["n"])])
        finally:
            # (an error between begin and end: the lists still in flight write into these buffers — end them,
            # whatever they return, before the buffers go)
            while lib.g2s_fill_in_flight(self.h) > 0:
                lib.g2s_fill_end(self.h)
            for c in ctx:
                if c["rbuf"] is not None:
                    c["rbuf"].free()
                    c["arena"].free()
        return out, t

    def fill_batch_onecall(self, gaps, pinned=False, want_timing=False):
        """g2s_fill_batch (prepare + run + free in one ABI call; lists longer than the group size
        go through the group pipeline, with the session's team when it has one)."""
        lib = load_library()
        arr, keep = _gap_array(gaps)
        nbytes = lib.g2s_team_arena_bytes(self.h, arr, len(gaps))
        bufs = []
        try:
            if pinned:
                bufs = [HostBuffer(max(1, nbytes)), HostBuffer(C.sizeof(g2s_result) * max(1, len(gaps)))]
                arena, res = bufs[0], bufs[1].array(g2s_result, max(1, len(gaps)))
                _check(lib.g2s_fill_batch(self.h, arr, len(gaps), res, C.cast(arena.p, C.c_char_p), nbytes))
            else:
                arena = C.create_string_buffer(max(1, nbytes))
                res = (g2s_result * max(1, len(gaps)))()
                _check(lib.g2s_fill_batch(self.h, arr, len(gaps), res, arena, nbytes))
            raw = arena.raw
            out = [FillResult(res[i], raw) for i in range(len(gaps))]
        finally:
            for hb in bufs:
                hb.free()
        if want_timing:
            t = g2s_timing()
            _check(lib.g2s_session_last_timing(self.h, C.byref(t)))
            return out, t
        return out

    def set_team(self, helpers, group_size=0):
        """g2s_session_set_team: fill_batch / execute_* on this session use self + helpers."""
        arr = (_VP * max(1, len(helpers)))(*[h.h for h in helpers])
        _check(load_library().g2s_session_set_team(self.h, arr, len(helpers), group_size))
        self._helpers = list(helpers)

    def prepare(self, gaps):
        """g2s_batch_prepare; returns an opaque PreparedBatch to run repeatedly."""
        return PreparedBatch(self, gaps)

    def execute_scaffolds(self, scaffolds_text, k, solid=2, max_fuz=10, nb_cores=1, max_mem_gb=20.0,
                          reads_label="reads.fa", filled_label="filled.fa"):
        lib = load_library()
        o = g2s_run_opts(k, solid, max_fuz, nb_cores, max_mem_gb)
        fa, lg = _VP(), _VP()
        gaps, filled = C.c_int32(0), C.c_int32(0)
        _check(lib.g2s_execute_scaffolds(self.h, C.byref(o), reads_label.encode(), filled_label.encode(),
                                         scaffolds_text.encode("ascii"), C.byref(fa), C.byref(lg), C.byref(gaps),
                                         C.byref(filled)))
        return _take_text(fa), _take_text(lg), gaps.value, filled.value

    def execute_scaffolds_stream(self, scaffolds_text, k, chunk_gaps, solid=2, max_fuz=10, nb_cores=1, max_mem_gb=20.0,
                                 reads_label="reads.fa", filled_label="filled.fa"):
        """g2s_execute_scaffolds_stream; returns (fasta pieces, log pieces, gaps, filled): one piece per batch."""
        lib = load_library()
        o = g2s_run_opts(k, solid, max_fuz, nb_cores, max_mem_gb)
        fa, lg = [], []
        on_fa = TEXT_FN(lambda t, n, u: fa.append(C.string_at(t, n).decode("ascii")))
        on_lg = TEXT_FN(lambda t, n, u: lg.append(C.string_at(t, n).decode("ascii")))
        gaps, filled = C.c_int32(0), C.c_int32(0)
        _check(lib.g2s_execute_scaffolds_stream(self.h, C.byref(o), reads_label.encode(), filled_label.encode(),
                                                scaffolds_text.encode("ascii"), chunk_gaps, on_fa, on_lg, None,
                                                C.byref(gaps), C.byref(filled)))
        return fa, lg, gaps.value, filled.value

    def execute_single(self, left, right, length, k, solid=2, max_fuz=10, max_mem_gb=20.0,
                       reads_label="reads.fa", filled_label="filled.fa"):
        lib = load_library()
        o = g2s_run_opts(k, solid, max_fuz, 1, max_mem_gb)
        fa, lg = _VP(), _VP()
        _check(lib.g2s_execute_single(self.h, C.byref(o), reads_label.encode(), filled_label.encode(),
                                      left.encode("ascii"), right.encode("ascii"), length, C.byref(fa), C.byref(lg)))
        return _take_text(fa), _take_text(lg)

    def destroy(self):
        if self.h:
            load_library().g2s_session_destroy(self.h)
            self.h = None


def team_fill(sessions, gaps, group_size=0, want_timing=False, prepared=None, pinned=False):
    """g2s_team_fill: the sessions (one or more per device) share one gap list; results equal
    sessions[0].fill_batch(gaps).  `prepared` = (arr, keep, arena, res) from a previous call to
    skip the marshalling (bench).  pinned: results and arena in g2s_host_alloc memory (every session's kernels then
    write its own group's results there: phase D3 sharded over the team); the buffers are not freed (tests)."""
    lib = load_library()
    if prepared is None:
        arr, keep = _gap_array(gaps)
        nbytes = lib.g2s_team_arena_bytes(sessions[0].h, arr, len(gaps))
        if pinned:
            arena = HostBuffer(max(1, nbytes))
            rbuf = HostBuffer(C.sizeof(g2s_result) * max(1, len(gaps)))
            res = rbuf.array(g2s_result, max(1, len(gaps)))
            hs = (_VP * len(sessions))(*[s.h for s in sessions])
            t = g2s_timing()
            try:
                _check(lib.g2s_team_fill(hs, len(sessions), arr, len(gaps), group_size, res, C.cast(arena.p, C.c_char_p), nbytes, C.byref(t)))
                raw = arena.raw
                out = [FillResult(res[i], raw) for i in range(len(gaps))]
            finally:
                arena.free()
                rbuf.free()
            return (out, t) if want_timing else out
        arena = C.create_string_buffer(max(1, nbytes))
        res = (g2s_result * max(1, len(gaps)))()
        prepared = (arr, keep, arena, res, nbytes)
    arr, keep, arena, res, nbytes = prepared
    hs = (_VP * len(sessions))(*[s.h for s in sessions])
    t = g2s_timing()
    _check(lib.g2s_team_fill(hs, len(sessions), arr, len(gaps), group_size, res, arena, nbytes, C.byref(t)))
    if want_timing == "raw":
        return prepared, t
    raw = arena.raw
    out = [FillResult(res[i], raw) for i in range(len(gaps))]
    return (out, t) if want_timing else out


class PreparedBatch:
    def __init__(self, session, gaps):
        lib = load_library()
        self.session = session
        self.n = len(gaps)
        self._arr, self._keep = _gap_array(gaps)
        self.h = _VP()
        _check(lib.g2s_batch_prepare(session.h, self._arr, self.n, C.byref(self.h)))
        self.nbytes = lib.g2s_batch_arena_bytes(self.h)
        self.arena = C.create_string_buffer(max(1, self.nbytes))
        self.res = (g2s_result * max(1, self.n))()

    def run(self):
        _check(load_library().g2s_batch_run(self.h, self.res, self.arena, self.nbytes))

    def timing(self):
        t = g2s_timing()
        _check(load_library().g2s_batch_timing(self.h, C.byref(t)))
        return t

    def results(self):
        raw = self.arena.raw
        return [FillResult(self.res[i], raw) for i in range(self.n)]

    def free(self):
        if self.h:
            load_library().g2s_batch_free(self.h)
            self.h = None


def test_post_gap(graph, params, gap, states, c_count, lengths, reached_j, final_d, seed, skip):
    """TEST HOOK binding (host half of phase D on a supplied DP table)."""
    lib = load_library()
    arr, keep = _gap_array([gap])
    n = len(states)
    nodes = (C.c_uint32 * max(1, n))(*[s[0] for s in states])
    depths = (C.c_int32 * max(1, n))(*[s[1] for s in states])
    counts = (C.c_uint32 * max(1, n))(*[s[2] for s in states])
    lens = (C.c_int32 * 2)(*(list(lengths) + [0, 0])[:2])
    res = g2s_result()
    buf = C.create_string_buffer(gap.gap_len + graph.k + params.d_err + gap.lmf + gap.rmf + 3)
    _check(lib.g2s_test_post_gap(graph.h, C.byref(params), arr, n, nodes, depths, counts, c_count, len(lengths), lens,
                                 reached_j, final_d, seed, skip, C.byref(res), buf))
    return FillResult(res, buf.raw)


def test_post_closure(graph, params, gap, records, xp, c_count, lengths, reached_j, final_d, seed, skip):
    """TEST HOOK binding: host D2 + D3 on a closure in the kernels' output layout.
    records: list of (node, count, meta, pred) with pred as a signed int."""
    lib = load_library()
    arr, keep = _gap_array([gap])
    n = len(records)
    flat = (C.c_uint32 * max(1, 4 * n))()
    for i, (node, cnt, meta, pred) in enumerate(records):
        flat[4 * i], flat[4 * i + 1], flat[4 * i + 2], flat[4 * i + 3] = node, cnt, meta, pred & 0xFFFFFFFF
    xs = (C.c_uint64 * max(1, len(xp)))(*xp)
    lens = (C.c_int32 * 2)(*(list(lengths) + [0, 0])[:2])
    res = g2s_result()
    buf = C.create_string_buffer(gap.gap_len + graph.k + params.d_err + gap.lmf + gap.rmf + 3)
    _check(lib.g2s_test_post_closure(graph.h, C.byref(params), arr, n, flat, len(xp), xs, c_count, len(lengths), lens,
                                     reached_j, final_d, seed, skip, C.byref(res), buf))
    return FillResult(res, buf.raw)


def test_post_segments(graph, params, gap, segs, c_count, lengths, reached_j, final_d, seed, skip):
    """TEST HOOK binding: host D2 + D3 directly on closure segments; returns (FillResult, on_segments)."""
    lib = load_library()
    arr, keep = _gap_array([gap])
    flat = (C.c_uint32 * max(1, 8 * len(segs)))()
    for i, rec in enumerate(segs):
        for q in range(8):
            flat[8 * i + q] = rec[q] & 0xFFFFFFFF
    lens = (C.c_int32 * 2)(*(list(lengths) + [0, 0])[:2])
    res = g2s_result()
    on = C.c_int32(0)
    buf = C.create_string_buffer(gap.gap_len + graph.k + params.d_err + gap.lmf + gap.rmf + 3)
    _check(lib.g2s_test_post_segments(graph.h, C.byref(params), arr, len(segs), flat, c_count, len(lengths), lens,
                                      reached_j, final_d, seed, skip, C.byref(res), buf, C.byref(on)))
    return FillResult(res, buf.raw), bool(on.value)


def test_seg_expand(graph, params, gap, segs, lengths, reached_j, n_records, n_xp):
    """TEST HOOK binding: closure segments (8 words each) -> ([(node, cnt, meta, pred)], sorted xp)."""
    lib = load_library()
    arr, keep = _gap_array([gap])
    flat = (C.c_uint32 * max(1, 8 * len(segs)))()
    for i, rec in enumerate(segs):
        for q in range(8):
            flat[8 * i + q] = rec[q] & 0xFFFFFFFF
    lens = (C.c_int32 * 2)(*(list(lengths) + [0, 0])[:2])
    out = (C.c_uint32 * max(1, 4 * n_records))()
    xs = (C.c_uint64 * max(1, n_xp))()
    _check(lib.g2s_test_seg_expand(graph.h, C.byref(params), arr, len(segs), flat, len(lengths), lens, reached_j,
                                   n_records, out, n_xp, xs))
    recs = [(out[4 * i], out[4 * i + 1], out[4 * i + 2], out[4 * i + 3] - (1 << 32) if out[4 * i + 3] >> 31 else out[4 * i + 3])
            for i in range(n_records)]
    return recs, sorted(xs[i] for i in range(n_xp))


def test_graph_tables(graph):
    """TEST HOOK binding: (succ, ustart) as flat lists of ints: successor table (2n x 4) and bitmap words."""
    n = graph.num_kmers
    succ = (C.c_uint32 * max(1, 8 * n))()
    words = (C.c_uint64 * max(1, (n + 63) // 64))()
    _check(load_library().g2s_test_graph_tables(graph.h, succ, words))
    return succ, words


def test_worker_pool(threads, rounds, n):
    """TEST HOOK binding: stress the host worker pool; raises G2SError on a lost or repeated task."""
    _check(load_library().g2s_test_worker_pool(threads, rounds, n))


def test_group_queue(nworkers, n, group_size):
    """TEST HOOK binding: the dispatcher's shared group counter with host threads in place of
    sessions; returns the worker each gap was handed to (raises G2SError on a lost or doubled gap)."""
    owner = (C.c_int32 * max(1, n))()
    _check(load_library().g2s_test_group_queue(nworkers, n, group_size, owner))
    return [owner[i] for i in range(n)]


def test_device_rand(device, seed, skip, n):
    """TEST HOOK binding: the same n values from the device's generator (g2s_rand_fill)."""
    out = (C.c_int32 * max(1, n))()
    _check(load_library().g2s_test_device_rand(device, seed, skip, n, out))
    return [out[i] for i in range(n)]


def test_group_queue_slow(nworkers, n, group_size, slow_worker, slow_us):
    """TEST HOOK binding: the group counter with one worker that is slow_us microseconds slower per group."""
    owner = (C.c_int32 * max(1, n))()
    _check(load_library().g2s_test_group_queue_slow(nworkers, n, group_size, slow_worker, slow_us, owner))
    return [owner[i] for i in range(n)]


def test_rand_skip(seed, skip, n):
    """TEST HOOK binding: the same values reached by a jump over `skip` values (g2s_share_end's way)."""
    out = (C.c_int32 * max(1, n))()
    _check(load_library().g2s_test_rand_skip(seed, skip, n, out))
    return [out[i] for i in range(n)]


def test_rand_stream(seed, skip, n):
    """TEST HOOK binding: n values of the product's rand() stream after srand(seed), skipping `skip`."""
    out = (C.c_int32 * max(1, n))()
    _check(load_library().g2s_test_rand_stream(seed, skip, n, out))
    return [out[i] for i in range(n)]
