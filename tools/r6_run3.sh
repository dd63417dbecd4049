set -x
O=gpurun_out/r6c
mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gputest.txt 2>&1
tail -4 $O/gputest.txt
for form in library allgather allreduce; do
  for i in 1 2; do
  CUGP_BCM_EXCHANGE=$form python bench.py --rehearse-rccl --experts-total 2 --rows 1500 --cpu-sample 0 --passes timed --steps 300 --warmup 30 2>> $O/bench.err | tail -1 > $O/rehearse_2x1500_${form}_$i.json
  done
  CUGP_BCM_EXCHANGE=$form python bench.py --rehearse-rccl --experts-total 1 --rows 6000 --cpu-sample 0 --passes timed --steps 40 --warmup 5 2>> $O/bench.err | tail -1 > $O/rehearse_1x6000_$form.json
  CUGP_BCM_EXCHANGE=$form python bench.py --rehearse-rccl --cpu-sample 0 --passes timed --sub-steps 0 2>> $O/bench.err | tail -1 > $O/rehearse_1x8192_$form.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6c/rehearse_*.json")):
    d=json.load(open(f)); print(f.split('/')[-1], round(d["ms_per_step"],4), {k:(round(v,4) if isinstance(v,float) else v) for k,v in d.get("exchange").items()})
PY
