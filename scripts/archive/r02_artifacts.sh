#!/bin/bash
# round-2 measurement artefacts (run on the GPU box through gpurun); everything lands in gpurun_out/
cd "$(dirname "$0")/../.."
O=gpurun_out
mkdir -p $O
python -m pytest tests -m gpu -q --durations=6 > $O/r02_gputests.log 2>&1
tail -4 $O/r02_gputests.log
python bench.py > $O/r02_bench_line.json 2> $O/r02_bench.err
for b in 32 16 8; do python bench.py --batch $b --no-cpu-baseline --no-general-extra 2>/dev/null > $O/r02_bench_batch$b.json; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof_r02 -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-general-extra > /root/repo/$O/r02_bench_under_rocprof.json 2> /root/repo/$O/prof_r02.err
cd /root/repo
f=$(find $O/prof_r02 -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && python scripts/summarize_rocprof.py "$f" $O/r02_bench_kernel_stats_summary.csv 30
python scripts/bench_configs.py c3 c4 c5 > $O/r02_configs.jsonl 2>/dev/null
cat $O/r02_bench_line.json | cut -c1-600
