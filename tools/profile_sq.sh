#!/bin/bash
# GPU box: per-gap cycle statistics (G2S_DUMP_STATS, tools/gapstats.py) and the SQ counter passes of the fill kernel
# on configs 2 and 3 — what profiles/r02_before_* / r02_mid_* were made with.  Results under gpurun_out/r02g.
mkdir -p gpurun_out/r02g; O=gpurun_out/r02g
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -f /tmp/st_c2.txt /tmp/st_c3.txt
G2S_DUMP_STATS=/tmp/st_c2.txt python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-c3-beside > /dev/null 2>&1; python tools/gapstats.py /tmp/st_c2.txt | tee $O/gapstats_c2.txt
G2S_DUMP_STATS=/tmp/st_c3.txt python bench.py --config C3 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1; python tools/gapstats.py /tmp/st_c3.txt | tee $O/gapstats_c3.txt
cp /tmp/st_c2.txt $O/; tail -10001 /tmp/st_c3.txt > $O/st_c3.txt
B="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-c3-beside"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/bench_under_rocprof.json 2> $O/rp1.err
grep g2s_fill_seg $O/stats/*/*_kernel_trace.csv | head -2 | cut -d, -f12-22
grep g2s_fill_seg $O/stats/*/*_kernel_stats.csv | sed 's/"g2s_fill_seg[^"]*"/g2s_fill_seg/' | cut -d, -f1-8
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS --output-format csv -d $O/sq1 -- $B > /dev/null 2> $O/rp4.err
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq2 -- $B > /dev/null 2> $O/rp5.err
python tools/pmc_sq_summary.py $O/pmc_sq_c2.json g2s_fill_seg $O/sq1 $O/sq2
B3="python3 bench.py --config C3 --steps 5 --warmup 1 --no-cpu-baseline"
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS --output-format csv -d $O/sq3 -- $B3 > /dev/null 2> $O/rp6.err
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq4 -- $B3 > /dev/null 2> $O/rp7.err
python tools/pmc_sq_summary.py $O/pmc_sq_c3.json g2s_fill_seg $O/sq3 $O/sq4
rm -rf $O/sq1 $O/sq2 $O/sq3 $O/sq4
python bench.py --config C3 --no-cpu-baseline | python tools/bsum.py C3
python bench.py --config C4 --no-cpu-baseline | python tools/bsum.py C4
python bench.py --config C5 --no-cpu-baseline | python tools/bsum.py C5
