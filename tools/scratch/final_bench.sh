python bench.py 2>&1 | tail -1 > gpurun_out/bench_default.json; cat gpurun_out/bench_default.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({k: d[k] for k in ('metric','value','unit','ms_per_step','n_gpus','scaling','dtype','vs_baseline')})); print(d['roofline']); print(d['cpu_baseline'])"
for n in 2 3; do
HSA_ENABLE_IPC_MODE_LEGACY=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2950$n bench.py --gpus $n --steps 2 --warmup 1 --dist-backend gloo --device 0 --checksum --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['value'], d['ms_per_step'], d.get('checksum'), d['config'])"
done
python bench.py --steps 2 --warmup 1 --checksum --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['value'], d['ms_per_step'], d.get('checksum'))"
