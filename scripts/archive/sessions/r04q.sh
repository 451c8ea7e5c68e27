#!/bin/bash
# round 4, GPU session Q: wider randomised sweeps on the final code (beyond the seeds in the suite)
cd "$(dirname "$0")/../.."
O=${1:-gpurun_out/r04q}; mkdir -p $O
export TMPDIR=/tmp
for seed in 3 4 5; do timeout 600 python scripts/solver_fuzz.py 100 $seed >> $O/solver_fuzz.jsonl 2>>$O/err.txt; done
for seed in 21 22 23; do timeout 600 python scripts/solver_fuzz_extensions.py $seed >> $O/solver_fuzz_ext.jsonl 2>>$O/err.txt; done
timeout 600 python scripts/grad_fuzz.py >> $O/grad_fuzz.jsonl 2>>$O/err.txt
grep -h summary $O/solver_fuzz.jsonl $O/solver_fuzz_ext.jsonl; tail -2 $O/grad_fuzz.jsonl | cut -c1-300
grep -vh summary $O/solver_fuzz.jsonl $O/solver_fuzz_ext.jsonl | cut -c1-330 | head -20
tail -3 $O/err.txt
