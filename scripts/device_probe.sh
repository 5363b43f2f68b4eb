#!/bin/bash
# Runs ON THE GPU BOX: what the device does while the two kinds of pass run (DESIGN 5b "When": devices of the pool differ, the split
# pass's GEMM tiles ran 35 % slower on one of them in round 4).  rocm-smi is sampled every 0.25 s beside (a) chunked passes, (b) split
# passes; the feed-forward launch's own stamps (PREGO_SPLIT_STATS, debug library) give the XCDs' shader clock from inside the kernel
# (s_memtime cycles per 10 ns s_memrealtime tick).  -> gpurun_out/device_probe/
O=$GRAFT_REPO_ROOT/gpurun_out/device_probe
mkdir -p $O
cd $GRAFT_REPO_ROOT
rocm-smi --showproductname --showclocks --showpower --showtemp > $O/idle.txt 2>&1
for M in 0 3; do
  ( while true; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (edge|junction)" | tr '\n' ' '; echo; sleep 0.25; done ) > $O/smi_pass$M.txt &
  SMI=$!
  PREGO_SPLIT_PASS=$M python3 bench.py --no-cpu-baseline --no-secondary --no-zero-flow --steps 40 --warmup 3 2>&1 | tail -1 | cut -c1-400 > $O/bench_pass$M.txt
  kill $SMI
done
PREGO_AMD_DEBUG_LIB=1 PREGO_SPLIT_STATS=1 PREGO_SPLIT_PASS=3 python3 scripts/probes/split_check.py 182 3342 22000 1 3 > $O/split_job_stats.log 2>&1
tail -12 $O/split_job_stats.log
for M in 0 3; do echo "== pass mode $M"; cat $O/bench_pass$M.txt | cut -c1-200; sort $O/smi_pass$M.txt | uniq -c | sort -rn | head -6; done
