for r in 1 2; do
for e in "X=1" "G2S_TRACE_IN_FILL=0" "G2S_FLANK_KERNEL=1" "G2S_FLANK_KERNEL=1 G2S_TRACE_IN_FILL=0 G2S_NO_EARLY_HANDOVER=1"; do
env $e python bench.py --no-cpu-baseline --config C3 --steps 30 | python tools/bsum.py "[$e]" | sed "s/gaps\/s.*| ms\/step/ms\/step/; s/kernel g2s_fill_seg //; s/+segx.*host us/host us/"
done; done
