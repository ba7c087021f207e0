#!/bin/bash
# Run on the MI355X box from the repo root: everything profiles/r06_* are copied from — the bench lines, rocprofv3
# kernel statistics and counter passes (FETCH / WRITE and SQ, each in a pass of its own; the program itself behind `--`),
# timelines of a step, the other workloads, streams of lists, N sessions / N ranks on the one device, the race hunt.
# Results under gpurun_out/$1.   usage: bash tools/measure_r06.sh [TAG] [race-hunt runs per build and path]
V=${1:-r06m}; RH=${2:-150}
O=gpurun_out/$V
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err < /dev/null
python tools/bsum.py C2-full < $O/bench.json
# the driver's own command line, three times (20 timed steps: a noisier line than the default 200)
for r in 1 2 3; do timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 < /dev/null | tee -a $O/driver_style.jsonl | python tools/bsum.py driver-style; done
# the slowest gaps of the two lists in the regular tier's kernel (the kernel's own clock reads)
timeout 300 python tools/slow_gaps.py C2 12 > $O/slowest_gaps_c2.txt 2>&1; head -3 $O/slowest_gaps_c2.txt | cut -c1-200
timeout 300 python tools/slow_gaps.py C3 25 > $O/slowest_gaps_c3.txt 2>&1; head -3 $O/slowest_gaps_c3.txt | cut -c1-200
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS"
SQ2="SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU"
for C in C2 C3 C5; do
  c=$(echo $C | tr A-Z a-z)
  ST=10; [ $C = C5 ] && ST=4
  B="python3 bench.py --config $C --steps $ST --warmup 2 --no-cpu-baseline --no-c3-beside"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$c -- $B > $O/bench_${c}_under_rocprof.json 2> $O/rp_$c.err
  cp $O/stats_$c/*/*_kernel_stats.csv $O/${c}_kernel_stats.csv; rm -rf $O/stats_$c
  python3 tools/kstats.py $O/${c}_kernel_stats.csv | head -14
  K=g2s_fill_seg2; [ $C = C3 ] && K="g2s_fill_seg("; [ $C = C5 ] && K=g2s_fill_segw
  K2=g2s_fill_seg2; [ $C = C3 ] && K2=g2s_fill_seg; [ $C = C5 ] && K2=g2s_fill_segw
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$c -- $B > /dev/null 2> $O/rpf_$c.err
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_$c -- $B > /dev/null 2> $O/rpw_$c.err
  python tools/pmc_summary.py $O/fetch_$c $O/write_$c "$K" $O/pmc_$c.json
  timeout 600 rocprofv3 --pmc $SQ1 --output-format csv -d $O/sq1_$c -- $B > /dev/null 2> $O/rps1_$c.err
  timeout 600 rocprofv3 --pmc $SQ2 --output-format csv -d $O/sq2_$c -- $B > /dev/null 2> $O/rps2_$c.err
  python tools/pmc_sq_summary.py $O/pmc_sq_$c.json $K2 $O/sq1_$c $O/sq2_$c
  [ $C = C3 ] && python tools/pmc_sq_summary.py $O/pmc_sq_c3_g2s_d2_small4.json g2s_d2_small4 $O/sq1_$c $O/sq2_$c
  rm -rf $O/fetch_$c $O/write_$c $O/sq1_$c $O/sq2_$c
done
# config 5 with phase D2 on the device: kernel statistics, SQ passes of g2s_d2_small4 / g2s_d2_big, per-closure log
B="python3 bench.py --config C5 --steps 4 --warmup 2 --no-cpu-baseline --no-c3-beside"
G2S_DEVICE_D2=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5d2 -- $B > $O/bench_c5_device_d2_under_rocprof.json 2> $O/rp_c5d2.err
cp $O/stats_c5d2/*/*_kernel_stats.csv $O/c5_device_d2_kernel_stats.csv; rm -rf $O/stats_c5d2
python3 tools/kstats.py $O/c5_device_d2_kernel_stats.csv | head -8
G2S_DEVICE_D2=1 timeout 600 rocprofv3 --pmc $SQ1 --output-format csv -d $O/sq1_c5d2 -- $B > /dev/null 2> $O/rps1_c5d2.err
G2S_DEVICE_D2=1 timeout 600 rocprofv3 --pmc $SQ2 --output-format csv -d $O/sq2_c5d2 -- $B > /dev/null 2> $O/rps2_c5d2.err
for K in g2s_d2_small4 g2s_d2_big; do python tools/pmc_sq_summary.py $O/pmc_sq_c5_$K.json $K $O/sq1_c5d2 $O/sq2_c5d2; done
rm -rf $O/sq1_c5d2 $O/sq2_c5d2
for C in C3 C5; do G2S_DEVICE_D2=1 G2S_D2_PROF=1 G2S_D2_LOG=$O/d2log_$C.txt timeout 600 python bench.py --config $C --no-cpu-baseline --no-c3-beside --steps 4 --warmup 1 --prime-seconds 0 2>&1 > /dev/null | grep "g2s_d2" | tail -2 | sed "s/^/$C: /" | tee -a $O/d2_sections.txt; done
{ echo "# config 3"; python tools/d2_log.py $O/d2log_C3.txt --top 3; echo "# config 5"; python tools/d2_log.py $O/d2log_C5.txt --top 6; } > $O/d2_closures.txt 2>&1; grep "closures;\|the last" $O/d2_closures.txt
# timelines of a step
bash tools/timeline.sh $V/tl_c2 C2 0.8 0.6 > /dev/null 2>&1; cp $O/tl_c2/timeline.txt $O/timeline_c2_step.txt
bash tools/timeline.sh $V/tl_c3 C3 2.6 1.8 > /dev/null 2>&1; cp $O/tl_c3/timeline.txt $O/timeline_c3_step.txt
rm -rf $O/tl_c2 $O/tl_c3
# the other workloads
for v in 0 1 2; do timeout 100 python bench.py --no-cpu-baseline --no-c3-beside --variant $v | tee -a $O/other.jsonl | python tools/bsum.py V$v; done
timeout 100 python bench.py --no-cpu-baseline --config C3 --gaps 1250 --steps 100 | tee -a $O/other.jsonl | python tools/bsum.py C3-1250
for r in 1 2 3; do
  G2S_DEVICE_D2=0 timeout 200 python bench.py --no-cpu-baseline --config C3 | tee -a $O/other.jsonl | python tools/bsum.py C3-10k-closures-on-the-host
  timeout 200 python bench.py --no-cpu-baseline --config C3 | tee -a $O/other.jsonl | python tools/bsum.py C3-10k
done
timeout 200 python bench.py --no-cpu-baseline --config C3 --variant 0 | tee -a $O/other.jsonl | python tools/bsum.py C3-10k-V0
G2S_TRACE_IN_FILL=0 timeout 200 python bench.py --no-cpu-baseline --config C3 --variant 0 | tee -a $O/other.jsonl | python tools/bsum.py C3-10k-V0-all-traced-by-phase-D3
timeout 400 python bench.py --no-cpu-baseline --config C4 | tee -a $O/other.jsonl | python tools/bsum.py C4
for r in 1 2; do timeout 400 python bench.py --no-cpu-baseline --config C5 --steps 5 | tee -a $O/other.jsonl | python tools/bsum.py C5; done
G2S_DEVICE_D2=1 timeout 400 python bench.py --no-cpu-baseline --config C5 --steps 5 | tee -a $O/other.jsonl | python tools/bsum.py C5-phase-D2-on-the-device
for r in 1 2; do timeout 400 python bench.py --config C3 --no-cpu-baseline --stream-lists 12 --steps 10 | tee -a $O/stream.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C3 stream', d['stream_lists'])"; done
timeout 400 python bench.py --no-cpu-baseline --no-c3-beside --stream-lists 10 --steps 50 | tee -a $O/stream.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2 stream', d['stream_lists'])"
timeout 200 python bench.py --no-cpu-baseline --no-c3-beside --prime-seconds 0 --warmup 0 --steps 20 | tee $O/cold.json | python tools/bsum.py C2-unprimed
# N sessions of one process / N ranks of a launcher, all on the one device (not a scaling curve)
{ bash tools/scale_shared.sh < /dev/null 2>&1 | cut -c1-900
  for N in 2 4; do
    timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600 + N)) bench.py --gpus $N --rank-per-gpu --share-device --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith(chr(123))][0])
print('one rank per GPU, N=$N ranks on the one device: strong', d['value'], 'gaps/s', d['ms_per_step'], 'ms/step | kernel ms by rank', d['roofline']['kernel_ms_per_launch_by_rank'], '|', d['config']['parallelism'][:160])"
  done
  timeout 600 python bench.py --gpus 2 --share-device --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('the default N=2 line (sessions on the one device): strong', d['value'], d['ms_per_step'], '| beside it:', json.dumps(d.get('weak_and_stream_beside'))[:900])"
} | tee $O/shared.txt
# every kernel path through the product build and the two race-hunting builds
bash tools/race_hunt.sh $V/race $RH > /dev/null 2>&1; cp $O/race/race_hunt.txt $O/race_hunt.txt; cat $O/race_hunt.txt
G2S_LIBRARY=$PWD/gap2seq_amd/_jit/libg2s_hip.so timeout 200 python bench.py --no-cpu-baseline --no-c3-beside --steps 50 | python tools/bsum.py "the jitter build on config 2 (its sleeps are real: compare the product build's line)" | tee -a $O/race_hunt.txt
