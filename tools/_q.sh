mkdir -p gpurun_out/q
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
PM_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 3 --no-cpu-baseline > gpurun_out/q/gloo2.json 2> gpurun_out/q/gloo2.err
tail -c 3000 gpurun_out/q/gloo2.json | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['n_gpus'], j['config']['row_tiles_per_rank'], {k: j['dp'][k] for k in ('allreduce_checksum_ok','params_in_sync_after_run','overlapped_fraction') if k in j['dp']}); print(j['dp'].get('buckets_timeline'))"
