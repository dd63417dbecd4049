set -x
O=gpurun_out/r6d
mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -s > $O/gputest.txt 2>&1
tail -4 $O/gputest.txt
grep -n "exp in k_build" $O/gputest.txt
python bench.py --sub-steps 0 --cpu-sample 0 2> $O/bench.err | tail -1 > $O/bench_line.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r6d/bench_line.json"))
print({k:d[k] for k in ("value","ms_per_step","cholesky_gflops")}, d["roofline_kbuild"]["launch_us"], d["roofline_kbuild"]["frac"], d["phase_ms_last"])
PY
python tools/ab2.py 8192 base 12=0 > $O/ab_exp_8192.txt 2>&1; cat $O/ab_exp_8192.txt
python tools/ab2.py 1500 base 12=0 > $O/ab_exp_1500.txt 2>&1; cat $O/ab_exp_1500.txt
python tools/bcm_ab.py > $O/bcm_one_gpu.txt 2>&1; cat $O/bcm_one_gpu.txt
python tools/ll_only_probe.py > $O/ll_only.txt 2>&1; cat $O/ll_only.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl_e1500
TL_EXPERTS=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_e1500 -- python3 $GRAFT_REPO_ROOT/tools/timeline_run.py 1500 > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/timeline_report.py $(find /tmp/tl_e1500 -name '*kernel_trace.csv' | head -1) --launches > $GRAFT_REPO_ROOT/$O/timeline_expert1500.txt
tail -25 $GRAFT_REPO_ROOT/$O/timeline_expert1500.txt
