#!/bin/bash
# round-2 (second half) measurement artefacts, run on the GPU box through gpurun; everything lands in gpurun_out/
cd "$(dirname "$0")/../.."
O=gpurun_out
mkdir -p $O
python -m pytest tests -m gpu -q --durations=6 > $O/r02b_gputests.log 2>&1
tail -3 $O/r02b_gputests.log
python bench.py > $O/r02b_bench_line.json 2> $O/r02b_bench.err
(cd scripts/micro && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/sp stream_patterns.hip && { /tmp/sp 32 16384; /tmp/sp 32 16384 symm; /tmp/sp 32 16384 occ; }) > $O/r02b_stream_patterns.jsonl 2>/dev/null
{ python scripts/symm_ab.py 32 16384 6; python scripts/symm_ab.py 64 16384 6; python scripts/symm_ab.py 32 16384 6 f32; } 2>/dev/null > $O/r02b_k1s_variants.jsonl
for b in 32 16 8; do python bench.py --batch $b --no-cpu-baseline --no-general-extra 2>/dev/null > $O/r02b_bench_batch$b.json; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof_r02b -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-general-extra > /root/repo/$O/r02b_bench_under_rocprof.json 2> /root/repo/$O/prof_r02b.err
cd /root/repo
f=$(find $O/prof_r02b -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && python scripts/summarize_rocprof.py "$f" $O/r02b_bench_kernel_stats_summary.csv 30
cut -c1-700 $O/r02b_bench_line.json
