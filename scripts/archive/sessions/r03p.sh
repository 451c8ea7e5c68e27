#!/bin/bash
# round 3, GPU session P: K3m artefacts — full sweep (order x batch x W x threads, both dtypes), phases of the final kernel,
# S2 un-restarted by variant, its timeline, the c2 lines (cold + warm call)
cd "$(dirname "$0")/../.."
O=gpurun_out/r03p; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python scripts/k3m_sweep.py 2>$O/sweep.err > $O/r03_k3m_sweep.jsonl
tail -4 $O/r03_k3m_sweep.jsonl | cut -c1-300
timeout 600 python scripts/s2_k3_variants.py 0:64 0:64 2:64 8:64 0:96 0:128 -1:64 0:64:False 2>/dev/null | grep -v amdgpu > $O/r03_s2_k3_variants.jsonl
cat $O/r03_s2_k3_variants.jsonl
timeout 300 python scripts/s2_timeline.py 2>/dev/null | grep -v amdgpu > $O/r03_s2_timeline.json
timeout 300 python scripts/bench_configs.py c2:S2:0 2>/dev/null > $O/r03_c2_S2.jsonl
timeout 300 python scripts/bench_configs.py c2:S2:96 2>/dev/null >> $O/r03_c2_S2.jsonl
timeout 300 python scripts/bench_configs.py c2:S3:0 2>/dev/null >> $O/r03_c2_S2.jsonl
cut -c1-500 $O/r03_c2_S2.jsonl
