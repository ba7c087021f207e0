for N in 1 2 4 8; do
  if [ $N -eq 1 ]; then timeout 200 python bench.py --no-cpu-baseline --steps 30 | python tools/bsum.py N1;
  else timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500+N)) bench.py --gpus $N --steps 30 --warmup 3 --backend gloo --share-device --no-cpu-baseline 2>/dev/null | python tools/bsum.py N$N; fi
done
