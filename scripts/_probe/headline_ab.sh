for rep in 1 2 3; do
for lib in "" scripts/_probe/lib_r06m.so; do
  XITORCH_AMD_LIB=$lib timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-general-extra --no-configs --no-standalone 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib [$lib]', round(d['ms_per_step'], 2), round(d['roofline'].get('avg_launch_ms') or 0, 4))"
done; done
