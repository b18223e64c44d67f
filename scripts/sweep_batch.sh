# wall-clock only (no per-kernel events): batch-size sweep of bench.py
for b in 2 3 4 6 8 16 33 0; do
  TDOA_BENCH_NOPROF=1 timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --batch $b 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d['value'], d['ms_per_step'])" $b
done
timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prof', d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])"
