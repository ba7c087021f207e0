#!/bin/bash
# The host half of the library (graph, FASTA, phase D on closures, execute, GapCutter/GapMerger records, the BAM reader and read filter) under
# AddressSanitizer, CPU build only (the GPU pool refuses sanitizer runs): the host objects are rebuilt with
# -fsanitize=address, linked with the kernels' objects as they are, and the CPU test files that drive that code
# through the library's test hooks run against it.  The in-tree library is put back afterwards.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
cd "$ROOT/gap2seq_amd/csrc"
make > /dev/null
for f in dbg fastx post g2s_execute synth gapio bam readfilter; do
  /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC -fsanitize=address -fno-omit-frame-pointer -x c++ -pthread -D__HIP_PLATFORM_AMD__ \
    -I/opt/rocm/include -c $f.cpp -o $T/$f.o
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -pthread -fsanitize=address -shared-libsan -o $T/libg2s_hip.so $T/*.o ../_build/*.hip.o -lz
cp ../libg2s_hip.so $T/normal.so
trap 'cp $T/normal.so "$ROOT/gap2seq_amd/libg2s_hip.so"; rm -rf $T' EXIT
cp $T/libg2s_hip.so ../libg2s_hip.so
cd "$ROOT"
RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 LD_PRELOAD=$RT python -m pytest tests/test_seg_model.py tests/test_gapio.py tests/test_readfilter.py tests/test_host.py -x -q
