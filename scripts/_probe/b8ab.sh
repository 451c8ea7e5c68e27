for rep in 1 2 3; do
for v in "" "--k1s-opts 16 --k1-streams 2" "--k1s-opts 16 --k1-streams 1"; do
  timeout 600 python bench.py --batch 8 --steps 30 --warmup 5 --no-cpu-baseline --no-general-extra --no-configs --no-standalone $v 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant [$v]', round(d['ms_per_step'], 3), round(d['roofline'].get('avg_launch_ms') or 0, 4))"
done; done
