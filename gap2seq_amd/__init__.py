"""gap2seq_amd — MI355X-native implementation of Gap2Seq-core's fill path.

The product is the C ABI in ``include/g2s.h`` (``gap2seq_amd/libg2s_hip.so``: host
graph builder in C++, HIP kernels for gfx950, host phase D) and the drop-in
``gap2seq_amd/Gap2Seq-core`` command line.  This Python package is only a thin
ctypes binding over that ABI for tests and ``bench.py``; it contains no compute
and no CPU fallback.
"""
from .lib import (G2S, G2SError, Graph, Session, Gap, load_library, library_path, team_fill,  # noqa: F401
                  G2S_GAP_SKIPPED, G2S_GAP_Q7, G2S_GAP_MEM_EXCEEDED, G2S_GAP_BACKTRACE_FAIL,
                  G2S_GAP_BAD_FLANK, G2S_GAP_PHASE_D)

__all__ = ["G2S", "G2SError", "Graph", "Session", "Gap", "load_library", "library_path", "team_fill"]
