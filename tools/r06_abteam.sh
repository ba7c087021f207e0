for r in 1 2; do
for e in "X=1" "G2S_D2_SMALL_WAVES=1" "G2S_DEVICE_D2=0"; do
env $e python bench.py --gpus 2 --share-device --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('N=2 shared $e', d['value'], d['ms_per_step'])"
env $e python bench.py --no-cpu-baseline --config C3 --steps 30 | python tools/bsum.py "C3 [$e]" | sed "s/gaps\/s.*| ms\/step/ms\/step/; s/kernel g2s_fill_seg //; s/+segx.*host us/host us/"
done; done
