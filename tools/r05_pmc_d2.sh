#!/bin/bash
# GPU box: SQ counter passes (rocprofv3 --pmc, no trace options) of g2s_d2_small / g2s_d2_big on config 5 with phase D2
# on the device, and of g2s_d2_small on config 3
O=gpurun_out/${1:-r05pmcd2}; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
export G2S_DEVICE_D2=1
for C in C5 C3; do
  c=$(echo $C | tr A-Z a-z)
  ST=4; [ $C = C3 ] && ST=10
  B="python3 bench.py --config $C --steps $ST --warmup 2 --no-cpu-baseline --no-c3-beside"
  timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS --output-format csv -d $O/sq1_$c -- $B > /dev/null 2> $O/rps1_$c.err
  timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq2_$c -- $B > /dev/null 2> $O/rps2_$c.err
  for K in g2s_d2_small g2s_d2_big; do
    [ $C = C3 ] && [ $K = g2s_d2_big ] && continue
    echo "== $C $K"; python tools/pmc_sq_summary.py $O/pmc_sq_${c}_$K.json $K $O/sq1_$c $O/sq2_$c
  done
  rm -rf $O/sq1_$c $O/sq2_$c
done
