#!/bin/bash
# Run on the MI355X box from the repo root: the bench line, the rocprofv3 stats and PMC passes
# that back it, and the other workloads of DESIGN.md §5.  Results under gpurun_out/$1.
# (rocprofv3 gets the program itself after `--`; counter passes carry no trace options.)
V=${1:-run}
WHAT=${2:-all}
O=gpurun_out/$V
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
B="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-c3-beside"
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err < /dev/null
python tools/bsum.py C2-full < $O/bench.json
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/bench_under_rocprof.json 2> $O/rp1.err
grep g2s_fill_seg $O/stats/*/*_kernel_stats.csv | sed 's/"\(g2s_fill_seg[a-z0-9]*\)[^"]*"/\1/' | cut -d, -f1-8
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B > /dev/null 2> $O/rp2.err
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $B > /dev/null 2> $O/rp3.err
python tools/pmc_summary.py $O/fetch $O/write g2s_fill_seg2 $O/pmc.json
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS --output-format csv -d $O/sq1 -- $B > /dev/null 2> $O/rp4.err
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq2 -- $B > /dev/null 2> $O/rp5.err
python tools/pmc_sq_summary.py $O/pmc_sq.json g2s_fill_seg2 $O/sq1 $O/sq2
for f in $O/rp4.err $O/rp5.err; do tail -n 2 $f | cut -c1-300; done
[ "$WHAT" = "c2" ] && exit 0
for v in 0 1 2; do timeout 100 python bench.py --no-cpu-baseline --no-c3-beside --variant $v | tee -a $O/variants.json | python tools/bsum.py V$v; done
timeout 100 python bench.py --no-cpu-baseline --config C3 --gaps 1250 --steps 100 | tee -a $O/c3.json | python tools/bsum.py C3-1250
timeout 200 python bench.py --no-cpu-baseline --config C3 | tee -a $O/c3.json | python tools/bsum.py C3-10k
timeout 400 python bench.py --no-cpu-baseline --config C4 | tee $O/c4.json | python tools/bsum.py C4
timeout 400 python bench.py --no-cpu-baseline --config C5 | tee $O/c5.json | python tools/bsum.py C5
# the deep-gap configuration: both kernels of the segment tier, FETCH_SIZE / WRITE_SIZE of the large variant
BC5="python3 bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5 -- $BC5 > $O/c5_under_rocprof.json 2> $O/rp6.err
grep g2s_fill_seg $O/stats_c5/*/*_kernel_stats.csv | sed 's/"\(g2s_fill_seg[a-z0-9]*\)[^"]*"/\1/' | cut -d, -f1-8
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_c5 -- $BC5 > /dev/null 2> $O/rp7.err
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_c5 -- $BC5 > /dev/null 2> $O/rp8.err
python tools/pmc_summary.py $O/fetch_c5 $O/write_c5 g2s_fill_segx $O/pmc_c5.json
bash tools/scale_shared.sh < /dev/null | tee $O/shared.txt
