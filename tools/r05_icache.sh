#!/bin/bash
# GPU box: g2s_d2_small on config 5 with 128 and with 1024 workgroups — instruction fetch counters (rocprofv3 --pmc)
O=gpurun_out/${1:-r05icache}; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 -L 2>/dev/null | grep -o -i "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQC_INST[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*\|SQ_INST_CYCLES[A-Z_]*" | sort -u | tr '\n' ' ' > $O/avail.txt; cat $O/avail.txt; echo
export G2S_DEVICE_D2=1
for W in 128 1024; do
  export G2S_D2_SMALL_WGS=$W
  B="python3 bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline --no-c3-beside --prime-seconds 0"
  timeout 400 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_IFETCH SQ_IFETCH_LEVEL --output-format csv -d $O/a_$W -- $B > /dev/null 2> $O/rpa_$W.err
  timeout 400 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/b_$W -- $B > /dev/null 2> $O/rpb_$W.err
  echo "== $W workgroups"; python tools/pmc_sq_summary.py $O/pmc_$W.json g2s_d2_small $O/a_$W $O/b_$W; tail -2 $O/rpa_$W.err $O/rpb_$W.err | grep -i "error\|invalid" | head -4
  rm -rf $O/a_$W $O/b_$W
done
