#!/bin/bash
# last validation of the round: full GPU suite, bench line (default run), torchrun forms of the bench
cd "$(dirname "$0")/../.."
O=gpurun_out
mkdir -p $O
python -m pytest tests -m gpu -q > $O/r02c_gputests.log 2>&1
tail -2 $O/r02c_gputests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > $O/r02c_bench_line.json 2> $O/r02c_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02c_bench_line.json')); r=d['roofline']
print('bench', round(d['value'],1), round(d['ms_per_step'],2), 'frac', round(r['frac'],4), 'stream', round(r['stream_read']['GBps'],1), 'of_stream', round(r['frac_of_stream_read'],4), 'standalone', round(r['standalone_whole_batch_launch']['frac'],4), 'general', round(d['general_k1']['value'],1), 'cpu', round(d['cpu_baseline']['value'],1))
PY
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-general-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('torchrun n=1', round(d['ms_per_step'],2), d['n_gpus'], d['config'].get('comm_backend'))"
XITORCH_BENCH_FORCE_PG=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-general-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('force_pg', round(d['ms_per_step'],2), d['config'].get('comm_backend'), d['config'].get('comm_world_size'))"
