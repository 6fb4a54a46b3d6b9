export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --device 0 --backend gloo --no-cpu-baseline --no-extras > $O/bench_torchrun2.json 2> $O/bench_torchrun2.err
tail -c 1500 $O/bench_torchrun2.json; echo; tail -3 $O/bench_torchrun2.err
python - <<'PY'
import json
p=json.load(open('gpurun_out/r4/bench_torchrun2.json'))
print(p['n_gpus'], p['ms_per_step'], p['value'], json.dumps(p['gradient_exchange'])[:600])
PY
