#!/bin/bash
# Run on the MI355X box from the repo root: the bench line, the rocprofv3 stats and PMC passes
# that back it, and the other workloads of DESIGN.md §5.  Results under gpurun_out/$1.
# (rocprofv3 gets the program itself after `--`; counter passes carry no trace options.)
V=${1:-run}
O=gpurun_out/$V
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rp1.err
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/rp2.err
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/rp3.err
python tools/pmc_summary.py $O/fetch $O/write g2s_fill_lds $O/pmc.json > /dev/null
python tools/bsum.py C2-full < $O/bench.json
python tools/bsum.py C2-rocprof < $O/bench_under_rocprof.json
grep g2s_fill_lds $O/stats/*/*_kernel_stats.csv | sed 's/"g2s_fill_lds[^"]*"/g2s_fill_lds/' | cut -d, -f1-8
for i in 1 2; do timeout 100 python bench.py --no-cpu-baseline | tee -a $O/c2_more.json | python tools/bsum.py C2; done
for v in 0 1 2; do timeout 100 python bench.py --no-cpu-baseline --variant $v | tee -a $O/variants.json | python tools/bsum.py V$v; done
timeout 100 python bench.py --no-cpu-baseline --gaps 1250 --steps 100 | tee -a $O/c3.json | python tools/bsum.py C3-1250
for i in 1 2; do timeout 200 python bench.py --no-cpu-baseline --gaps 10000 --steps 20 --warmup 3 | tee -a $O/c3.json | python tools/bsum.py C3-10k; done
timeout 400 python bench.py --no-cpu-baseline --genome 60000000 --k 63 --gaps 2000 --steps 30 --warmup 3 | tee $O/c4.json | python tools/bsum.py C4
timeout 400 python bench.py --no-cpu-baseline --gaps 1000 --min-len 2000 --max-len 5000 --dist-error 2000 --steps 3 --warmup 1 | tee $O/c5.json | python tools/bsum.py C5
