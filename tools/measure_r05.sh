#!/bin/bash
# Run on the MI355X box from the repo root: the bench lines, the rocprofv3 kernel statistics and counter passes that
# back them, the other workloads — what profiles/r05_* are copied from.  Results under gpurun_out/$1.
# (rocprofv3 gets the program itself after `--`; counter passes carry no trace options.)
V=${1:-r05m}
O=gpurun_out/$V
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err < /dev/null
python tools/bsum.py C2-full < $O/bench.json
for C in C2 C3 C5; do
  c=$(echo $C | tr A-Z a-z)
  ST=10; [ $C = C5 ] && ST=4
  B="python3 bench.py --config $C --steps $ST --warmup 2 --no-cpu-baseline --no-c3-beside"
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$c -- $B > $O/bench_${c}_under_rocprof.json 2> $O/rp_$c.err
  cp $O/stats_$c/*/*_kernel_stats.csv $O/${c}_kernel_stats.csv; rm -rf $O/stats_$c
  python3 tools/kstats.py $O/${c}_kernel_stats.csv | head -14
  [ $C = C5 ] && continue
  timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$c -- $B > /dev/null 2> $O/rpf_$c.err
  timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_$c -- $B > /dev/null 2> $O/rpw_$c.err
  K=g2s_fill_seg2; [ $C = C3 ] && K="g2s_fill_seg("
  python tools/pmc_summary.py $O/fetch_$c $O/write_$c "$K" $O/pmc_$c.json
  timeout 400 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS --output-format csv -d $O/sq1_$c -- $B > /dev/null 2> $O/rps1_$c.err
  timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq2_$c -- $B > /dev/null 2> $O/rps2_$c.err
  K2=g2s_fill_seg2; [ $C = C3 ] && K2=g2s_fill_seg
  python tools/pmc_sq_summary.py $O/pmc_sq_$c.json $K2 $O/sq1_$c $O/sq2_$c
  rm -rf $O/fetch_$c $O/write_$c $O/sq1_$c $O/sq2_$c
done
# config 5 with phase D2 on the device (G2S_DEVICE_D2=1): kernel statistics and the section profile of g2s_d2_*
B="python3 bench.py --config C5 --steps 4 --warmup 2 --no-cpu-baseline --no-c3-beside"
G2S_DEVICE_D2=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5d2 -- $B > $O/bench_c5_device_d2_under_rocprof.json 2> $O/rp_c5d2.err
cp $O/stats_c5d2/*/*_kernel_stats.csv $O/c5_device_d2_kernel_stats.csv; rm -rf $O/stats_c5d2
python3 tools/kstats.py $O/c5_device_d2_kernel_stats.csv | head -8
for C in C3 C5; do G2S_DEVICE_D2=1 G2S_D2_PROF=1 G2S_D2_LOG=$O/d2log_$C.txt timeout 600 python bench.py --config $C --no-cpu-baseline --no-c3-beside --steps 4 --warmup 1 --prime-seconds 0 2>&1 > /dev/null | grep "g2s_d2" | tail -2 | sed "s/^/$C: /" | tee -a $O/d2_sections.txt; done
{ echo "# config 3"; python tools/d2_log.py $O/d2log_C3.txt --top 3; echo "# config 5"; python tools/d2_log.py $O/d2log_C5.txt --top 6 --poll-model 32 16; } > $O/d2_closures.txt; grep "closures;\|the last" $O/d2_closures.txt
for v in 0 1 2; do timeout 100 python bench.py --no-cpu-baseline --no-c3-beside --variant $v | tee -a $O/other.jsonl | python tools/bsum.py V$v; done
timeout 100 python bench.py --no-cpu-baseline --config C3 --gaps 1250 --steps 100 | tee -a $O/other.jsonl | python tools/bsum.py C3-1250
for r in 1 2 3; do
  G2S_DEVICE_D2=0 timeout 200 python bench.py --no-cpu-baseline --config C3 | tee -a $O/other.jsonl | python tools/bsum.py C3-10k-closures-on-the-host
  timeout 200 python bench.py --no-cpu-baseline --config C3 | tee -a $O/other.jsonl | python tools/bsum.py C3-10k
done
timeout 400 python bench.py --no-cpu-baseline --config C4 | tee -a $O/other.jsonl | python tools/bsum.py C4
for r in 1 2; do timeout 400 python bench.py --no-cpu-baseline --config C5 --steps 5 | tee -a $O/other.jsonl | python tools/bsum.py C5; done
G2S_DEVICE_D2=1 timeout 400 python bench.py --no-cpu-baseline --config C5 --steps 5 | tee -a $O/other.jsonl | python tools/bsum.py C5-phase-D2-on-the-device
for r in 1 2; do timeout 400 python bench.py --config C3 --no-cpu-baseline --stream-lists 12 --steps 10 | tee -a $O/stream.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C3 stream', d['stream_lists'])"; done
timeout 400 python bench.py --no-cpu-baseline --no-c3-beside --stream-lists 10 --steps 50 | tee -a $O/stream.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2 stream', d['stream_lists'])"
timeout 200 python bench.py --no-cpu-baseline --no-c3-beside --prime-seconds 0 --warmup 0 --steps 20 | tee $O/cold.json | python tools/bsum.py C2-unprimed
bash tools/r05_scale_shared.sh < /dev/null 2>&1 | cut -c1-900 | tee $O/shared.txt
