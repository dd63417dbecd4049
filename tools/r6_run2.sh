set -x
O=gpurun_out/r6b
mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -s > $O/gputest.txt 2>&1
tail -4 $O/gputest.txt
grep -n "end point deviates\|probes," $O/gputest.txt
python bench.py --sub-steps 0 > $O/bench_line.json 2> $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r6b/bench_line.json"))
print({k:d[k] for k in ("value","ms_per_step","cholesky_gflops","library_build_id")}, d["roofline"]["traffic"], d["roofline"]["traffic_source"][:80], d["cpu_baseline"]["host_cpu"], d["cpu_baseline"]["host_cores_total"], d["cpu_baseline"]["pinned_core"], d.get("exchange"))
PY
for form in allgather allreduce; do
  CUGP_BCM_EXCHANGE=$form python bench.py --rehearse-rccl --experts-total 2 --rows 1500 --cpu-sample 0 --passes timed --steps 200 --warmup 20 > $O/rehearse_2x1500_$form.json 2>> $O/bench.err
  CUGP_BCM_EXCHANGE=$form python bench.py --rehearse-rccl --experts-total 1 --rows 6000 --cpu-sample 0 --passes timed --steps 40 --warmup 5 > $O/rehearse_1x6000_$form.json 2>> $O/bench.err
  CUGP_BCM_EXCHANGE=$form python bench.py --rehearse-rccl --cpu-sample 0 --passes timed --sub-steps 0 > $O/rehearse_1x8192_$form.json 2>> $O/bench.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6b/rehearse_*.json")):
    d=json.load(open(f)); print(f, round(d["ms_per_step"],4), d.get("exchange"))
PY
tools/bin/wgtimes 8192 0 > $O/wgtimes_8192_S1.txt 2>&1
tools/bin/wgtimes 8192 0 17=2 > $O/wgtimes_8192_S2.txt 2>&1
tools/bin/chain_bench > $O/chain_bench.txt 2>&1
tail -30 $O/chain_bench.txt
