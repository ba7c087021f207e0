mkdir -p gpurun_out/r02k; O=gpurun_out/r02k
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python bench.py --config C3 --no-cpu-baseline --steps 10 | python tools/bsum.py C3-dev
G2S_HOST_LOOKUP=1 python bench.py --config C3 --no-cpu-baseline --steps 10 | python tools/bsum.py C3-hostlookup
python bench.py --config C3 --no-cpu-baseline --steps 10 | python tools/bsum.py C3-dev
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --config C3 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/rp1.err
cat $O/stats/*/*_kernel_stats.csv | cut -c1-200 | head -12
