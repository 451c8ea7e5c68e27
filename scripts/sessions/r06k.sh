#!/bin/bash
# r06k: K3p from order 64 on in the Davidson driver: solver tests, configs[0], the headline (short), the 8-operator shard,
# the configs[4] shard — A/B against the K3t routing (XITORCH_K3P_MIN_K=129 restores it)
cd "$(dirname "$0")/../.."
O=gpurun_out/r06k; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_davidson.py tests/test_gpu_fuzz.py tests/test_gpu_dist.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests_tail.txt
python scripts/c1_profile.py 5 2>&1 | tail -1 | tee $O/c1_wall.json
for mk in 64 129; do
  echo "== K3P_MIN_K=$mk"
  XITORCH_K3P_MIN_K=$mk timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-general-extra --no-configs --no-standalone 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'ms_per_step': d['ms_per_step'], 'value': d['value'], 'avg_launch_ms': d['roofline'].get('avg_launch_ms')}))" | tee $O/headline_mk$mk.json
  XITORCH_K3P_MIN_K=$mk timeout 600 python bench.py --batch 8 --steps 10 --warmup 3 --no-cpu-baseline --no-general-extra --no-configs --no-standalone 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'b8_ms_per_step': d['ms_per_step']}))" | tee $O/b8_mk$mk.json
  XITORCH_K3P_MIN_K=$mk timeout 600 python bench.py --config c5w --steps 4 --warmup 2 --no-general-extra 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'c5w_ms_per_step': d['ms_per_step'], 'launch_ms': d['roofline'].get('avg_launch_ms')}))" | tee $O/c5w_mk$mk.json
done
exit 0
