#!/bin/bash
# round 3, GPU session A: new K1s (deterministic, runs) — correctness, A/B alone, A/B inside the eigensolver pipeline
cd "$(dirname "$0")/../.."
O=gpurun_out/r03a; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_gpu_k1.py -m gpu -x -q -k "symm" 2>&1 | tail -5 > $O/pytest_symm.txt
cat $O/pytest_symm.txt
{ python scripts/k1s_ab.py 32 16384 6; python scripts/k1s_ab.py 64 16384 6; python scripts/k1s_ab.py 16 32768 6 f32; } 2>$O/ab.err > $O/k1s_ab.jsonl
cat $O/k1s_ab.jsonl
for L in 1 2 4; do
  python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-general-extra --k1s-run $L 2>$O/bench_L$L.err | tee $O/bench_L$L.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('L=$L ms/step', round(d['ms_per_step'],2), 'k1s_ms', round(r['avg_launch_ms'],3), 'frac', round(r['frac'],4), 'ok', d['check']['ok'])"
done
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/pytest_all.txt
cat $O/pytest_all.txt
