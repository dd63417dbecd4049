O=$GRAFT_REPO_ROOT/gpurun_out/r6f
mkdir -p $O
python tools/exp_check.py
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  rm -rf $O/pmc_exp$v
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/pmc_exp$v -- python3 $GRAFT_REPO_ROOT/bench.py --passes timed --cpu-sample 0 --sub-steps 0 --steps 2 --warmup 1 --tune 12=$v > $O/pmc_line$v.json 2> $O/pmc_err$v.txt
  python3 - <<PY
import csv,glob,collections
f=glob.glob("$O/pmc_exp$v/*/*counter_collection.csv")[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter(); dur=collections.defaultdict(float); seen=set()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0].replace("cugp::","")
    if k in ("k_build","k_trace"):
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        if (r["Dispatch_Id"]) not in seen:
            seen.add(r["Dispatch_Id"]); n[k]+=1; dur[k]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))*1e-3
for k in agg:
    a=agg[k]; print("exp=$v", k, "launches", n[k], "avg us %.1f"%(dur[k]/n[k]), "VALU/wave %.0f"%(a["SQ_INSTS_VALU"]/a["SQ_WAVES"]), "SALU/wave %.0f"%(a["SQ_INSTS_SALU"]/a["SQ_WAVES"]), "wait_any %.2f wait_inst %.2f active %.2f"%(a["SQ_WAIT_ANY"]/a["SQ_WAVE_CYCLES"],a["SQ_WAIT_INST_ANY"]/a["SQ_WAVE_CYCLES"],a["SQ_ACTIVE_INST_ANY"]/a["SQ_WAVE_CYCLES"]))
PY
done
