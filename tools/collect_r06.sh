#!/bin/bash
# copies what tools/measure_r06.sh left under gpurun_out/$1 into profiles/ as r06_* (the tracked copies DESIGN.md cites)
S=gpurun_out/${1:-r06m}; P=profiles
cp $S/bench.json $P/r06_bench.json
cp $S/driver_style.jsonl $P/r06_bench_driver_style.jsonl; cp $S/slowest_gaps_c2.txt $P/r06_slowest_gaps_c2.txt; cp $S/slowest_gaps_c3.txt $P/r06_slowest_gaps_c3.txt
for c in c2 c3 c5; do cp $S/bench_${c}_under_rocprof.json $P/r06_bench_${c}_under_rocprof.json; cp $S/${c}_kernel_stats.csv $P/r06_${c}_kernel_stats.csv; cp $S/pmc_$c.json $P/r06_pmc_$c.json; cp $S/pmc_sq_$c.json $P/r06_pmc_sq_$c.json; done
cp $S/bench_c5_device_d2_under_rocprof.json $P/r06_bench_c5_device_d2_under_rocprof.json
cp $S/c5_device_d2_kernel_stats.csv $P/r06_c5_device_d2_kernel_stats.csv
for k in c3_g2s_d2_small4 c5_g2s_d2_small4 c5_g2s_d2_big; do cp $S/pmc_sq_$k.json $P/r06_pmc_sq_$k.json; done
cp $S/timeline_c2_step.txt $P/r06_timeline_c2_step.txt; cp $S/timeline_c3_step.txt $P/r06_timeline_c3_step.txt
cp $S/other.jsonl $P/r06_other_workloads.jsonl; cp $S/stream.jsonl $P/r06_stream_lists.jsonl; cp $S/cold.json $P/r06_cold_start.json
cp $S/shared.txt $P/r06_shared_device_sessions.txt
{ cat $S/race_hunt.txt; echo; echo "# per run (tools/race_hunt.py):"; cat $S/race/race_hunt.jsonl; } > $P/r06_race_hunt.txt
cp $S/d2_closures.txt $P/r06_d2_closures.txt; cp $S/d2_sections.txt $P/r06_d2_sections.txt
ls $P/r06_* | wc -l
