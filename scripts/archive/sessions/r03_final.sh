#!/bin/bash
# round 3: validation + artefacts of the final code (one gpurun call).  Copies of the summaries go to profiles/r03_*.
cd "$(dirname "$0")/../.."
O=gpurun_out/r03_final; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -q > $O/r03_gputests.log 2>&1
tail -3 $O/r03_gputests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > $O/r03_bench_line.json 2> $O/r03_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_final/r03_bench_line.json')); r=d['roofline']; c=d['cpu_baseline']
print('bench', round(d['value'],1), round(d['ms_per_step'],2), 'frac', round(r['frac'],4), 'k1s_ms', round(r['avg_launch_ms'],3), 'traffic', r['traffic'], 'stream', round(r['stream_read']['GBps'],1), 'standalone', round(r['standalone_whole_batch_launch']['frac'],4), 'general', round(d['general_k1']['value'],1), 'cpu', round(c['value'],1), c.get('config1_n512_b1'), c.get('k1_product_cpu'))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-general-extra > $O/r03_bench_under_rocprof.json 2> $O/prof.err
F=$(find $O/prof -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && python scripts/summarize_rocprof.py $F $O/r03_bench_kernel_stats_summary.csv 30 && head -8 $O/r03_bench_kernel_stats_summary.csv
rm -rf $O/prof
for b in 32 16 8; do python bench.py --batch $b --steps 5 --warmup 2 --no-cpu-baseline --no-general-extra 2>/dev/null; done > $O/r03_strong_scaling_shards.jsonl
python -c "
import json
for l in open('gpurun_out/r03_final/r03_strong_scaling_shards.jsonl'):
    d=json.loads(l); print('shard', d['config']['global_batch'], round(d['ms_per_step'],2))"
for b in 8 16 64; do python scripts/timeline_small.py $b overlap_only=1 2>/dev/null; done > $O/r03_timeline_shards.jsonl
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof8 -- python bench.py --batch 8 --steps 10 --warmup 3 --no-cpu-baseline --no-general-extra > /dev/null 2>$O/prof8.err
F=$(find $O/prof8 -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && python scripts/summarize_rocprof.py $F $O/r03_b8_kernel_stats_summary.csv 30
rm -rf $O/prof8
timeout 900 python scripts/bench_configs.py c3 c3g c4 c5 c5w 2>$O/configs.err > $O/r03_secondary_configs.jsonl
# (the hard-spectrum runs each in a process of their own: 137 GB of operators + growing work buffers)
timeout 300 python scripts/bench_configs.py c2:S2:96 2>>$O/configs.err >> $O/r03_secondary_configs.jsonl
timeout 300 python scripts/bench_configs.py c2:S2:0 2>>$O/configs.err >> $O/r03_secondary_configs.jsonl
cut -c1-400 $O/r03_secondary_configs.jsonl
timeout 300 python scripts/k3m_sweep.py quick 2>/dev/null > $O/r03_k3m_sweep_quick.jsonl
tail -3 $O/r03_k3m_sweep_quick.jsonl | cut -c1-200
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-general-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('torchrun n=1', round(d['ms_per_step'],2), d['n_gpus'], d['config'].get('comm_backend'))"
XITORCH_BENCH_FORCE_PG=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-general-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('force_pg', round(d['ms_per_step'],2), d['config'].get('comm_backend'), d['config'].get('comm_world_size'))"
