O=$GRAFT_REPO_ROOT/gpurun_out/r6g
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  rm -rf /tmp/pe_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe_$v -- python3 $GRAFT_REPO_ROOT/bench.py --passes timed --cpu-sample 0 --sub-steps 0 --steps 10 --tune 12=$v > $O/line_exp$v.json 2> $O/err_exp$v.txt
  cp $(find /tmp/pe_$v -name '*kernel_stats.csv' | head -1) $O/kernel_stats_exp$v.csv
  python3 - <<PY
import csv
for r in csv.DictReader(open("$O/kernel_stats_exp$v.csv")):
    n=r["Name"]
    if any(k in n for k in ("k_build","k_trace","k_cross")):
        print("exp=$v  %-10s calls %s avg %.1f us min %.1f" % (n.split("(")[0].replace("cugp::",""), r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
done
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/gputest_parity.txt 2>&1; tail -3 $O/gputest_parity.txt
