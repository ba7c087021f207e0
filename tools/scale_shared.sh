#!/bin/bash
# One MI355X standing in for N: `bench.py --gpus N --share-device` puts the dispatcher's N sessions
# on device 0 (what a one-GPU box can show of the N>1 path: the group queue, the gather on the lead's device,
# phase D3 over the whole list, the equality with the one-session result — not a scaling curve).
# groups are cut per DEVICE: N sessions on one device share one group (the default)
for N in 1 2 4 8; do
  timeout 300 python bench.py --gpus $N --config C3 --share-device --no-cpu-baseline --steps 10 --warmup 2 | python tools/bsum.py N$N
done
# one group per SESSION (what N devices would get), and four per session (whoever is free takes the next one)
for N in 2 4 8; do
  timeout 300 python bench.py --gpus $N --config C3 --share-device --no-cpu-baseline --steps 10 --warmup 2 --group $((10000 / N)) | python tools/bsum.py N$N-groups-of-$((10000 / N))
done
for N in 2 8; do
  timeout 300 python bench.py --gpus $N --config C3 --share-device --no-cpu-baseline --steps 10 --warmup 2 --group $((10000 / (4 * N))) | python tools/bsum.py N$N-groups-of-$((10000 / (4 * N)))
done
