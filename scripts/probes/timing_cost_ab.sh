for i in 1 2 3; do
for v in 0 1; do
if [ $v = 1 ]; then export PREGO_BENCH_NO_KERNEL_TIMING=1; else unset PREGO_BENCH_NO_KERNEL_TIMING; fi
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-zero-flow 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('notiming=$v', round(d['ms_per_step'],2))"
done; done
