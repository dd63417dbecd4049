#!/bin/bash
# Regenerate everything under profiles/ for one round on the GPU box:  tools/make_profiles.sh r04
# Outputs land in gpurun_out/profiles_<tag>/ (copy the ones to keep into profiles/).
tag=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/profiles_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
set -x
python3 $R/bench.py > $O/${tag}_bench_line.json 2> $O/bench.err || exit 1
# one rocprofv3 --kernel-trace --stats run PER PASS of bench.py (timed = the default path `value` is measured on,
# profiled = the same schedule with HIP events around the MFMA launches (what roofline* is computed from),
# isolated = overlap off): a kernel's AverageNs in the CSV of a pass is then that pass's figure, nothing mixed
for pass in timed profiled isolated; do
  rm -rf /tmp/prof_$pass
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$pass -- python3 $R/bench.py --passes $pass --cpu-sample 0 --sub-steps 0 > $O/${tag}_bench_${pass}_line_under_rocprof.json 2> $O/bench_$pass.err || exit 1
  cp $(find /tmp/prof_$pass -name '*kernel_stats.csv' | head -1) $O/${tag}_bench_${pass}_n8192_kernel_stats.csv
done
# BASELINE config 3's shape (siproper_10000: 79 tiles, 16 rows of identity padding): the bench line and the kernel statistics of its timed pass
python3 $R/bench.py --rows 10000 --sub-steps 0 --cpu-sample 0 --steps 10 > $O/${tag}_bench_n10000_line.json 2>> $O/bench.err
rm -rf /tmp/prof_n10000
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_n10000 -- python3 $R/bench.py --rows 10000 --passes timed --steps 10 --cpu-sample 0 --sub-steps 0 > $O/${tag}_bench_n10000_line_under_rocprof.json 2>> $O/bench.err || exit 1
cp $(find /tmp/prof_n10000 -name '*kernel_stats.csv' | head -1) $O/${tag}_bench_n10000_kernel_stats.csv
python3 $R/bench.py --rows 1500 --experts-total 16 --cpu-sample 0 > $O/${tag}_bench_bcm16_line.json 2>> $O/bench.err
python3 $R/bench.py --rows 6000 --experts-total 4 --cpu-sample 0 --steps 10 > $O/${tag}_bench_bcm4_line.json 2>> $O/bench.err
python3 $R/bench.py --experts-per-gpu 2 --cpu-sample 0 --steps 10 > $O/${tag}_bench_2x8192_line.json 2>> $O/bench.err
# one rank of the 8-GPU BCM runs behind a real (one-rank) RCCL communicator: the library's exchange against the two
# torch.distributed forms (device_ms / collective_ms / host time between evaluations, bench.py `exchange`)
for form in library allgather allreduce; do
  CUGP_BCM_EXCHANGE=$form python3 $R/bench.py --rehearse-rccl --experts-total 2 --rows 1500 --cpu-sample 0 --passes timed --steps 300 --warmup 30 2>> $O/bench.err | tail -1 > $O/${tag}_rehearse_2x1500_$form.json
  CUGP_BCM_EXCHANGE=$form python3 $R/bench.py --rehearse-rccl --experts-total 1 --rows 6000 --cpu-sample 0 --passes timed --steps 40 --warmup 5 2>> $O/bench.err | tail -1 > $O/${tag}_rehearse_1x6000_$form.json
  CUGP_BCM_EXCHANGE=$form python3 $R/bench.py --rehearse-rccl --cpu-sample 0 --passes timed --sub-steps 0 2>> $O/bench.err | tail -1 > $O/${tag}_rehearse_1x8192_$form.json
done
python3 $R/tools/la_bench.py > $O/${tag}_la_bench.txt 2>> $O/bench.err
python3 $R/tools/ll_only_probe.py > $O/${tag}_ll_only.txt 2>> $O/bench.err
python3 $R/tools/bcm_ab.py > $O/${tag}_bcm_one_gpu.txt 2>> $O/bench.err
for mode in overlap serial; do
  rm -rf /tmp/tl_$mode
  pipe=$([ $mode = overlap ] && echo -1 || echo 0)
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$mode -- python3 $R/tools/timeline_run.py 8192 $pipe > /dev/null 2>&1 || exit 1
  python3 $R/tools/timeline_report.py $(find /tmp/tl_$mode -name '*kernel_trace.csv' | head -1) --launches > $O/${tag}_timeline_${mode}_n8192.txt
done
# one 1500-row expert and the 16 x 1500 group (config 5's shape on one GPU)
for shape in "expert1500 1500 0" "bcm16x1500 24000 16"; do
  set -- $shape
  rm -rf /tmp/tl_$1
  TL_EXPERTS=$3 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$1 -- python3 $R/tools/timeline_run.py $2 > /dev/null 2>&1 || exit 1
  python3 $R/tools/timeline_report.py $(find /tmp/tl_$1 -name '*kernel_trace.csv' | head -1) --launches > $O/${tag}_timeline_$1.txt
done
# PMC passes (separate passes; counters only with --kernel-trace)
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VMEM_WR SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_WAVES"; do
  t=$(echo $pass | cut -d' ' -f1)
  rm -rf $R/gpurun_out/pmc_$t
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$t -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --sub-steps 0 --passes timed > $R/gpurun_out/pmc_$t.log 2>&1 || echo "pass $t failed"
done
cd $R && python3 tools/pmc_summary.py $R/gpurun_out $tag && cp profiles/${tag}_pmc_summary.json $O/
# HBM-side bytes of a whole evaluation (the run above holds 3 + 1 evaluations: warm-up, 2 steps, 1 in front of predict)
python3 tools/pmc_total.py $R/gpurun_out 4 > $O/${tag}_pmc_total_traffic.txt 2>&1
# per-workgroup start / end / shader-clock stamps of the chain and tile kernels (diagnostic build, see tools/wgtimes.hip)
[ -x $R/tools/bin/wgtimes ] && $R/tools/bin/wgtimes 8192 1 > $O/${tag}_wgtimes_n8192.txt 2>&1
[ -x $R/tools/bin/wgtimes ] && $R/tools/bin/wgtimes 1500 1 > $O/${tag}_wgtimes_n1500.txt 2>&1
# board power and clocks while the metric workload loops (the evaluation is clock-limited: DESIGN section 8)
( python3 $R/bench.py --passes timed --steps 400 --cpu-sample 0 --sub-steps 0 > /dev/null 2>&1 & BP=$!; sleep 4; for i in 1 2 3 4 5; do rocm-smi --showpower --showclocks 2>/dev/null | grep -i "power\|sclk\|mclk"; sleep 0.5; done; wait $BP ) > $O/${tag}_power_clocks_under_load.txt 2>&1
# the factorisation's chain kernels alone, with in-kernel cycle stamps (built in the build container: see the sources' headers)
[ -x $R/tools/bin/chain_bench ] && $R/tools/bin/chain_bench > $O/${tag}_chain_bench.txt 2>&1
[ -x $R/tools/bin/panel_bench ] && $R/tools/bin/panel_bench >> $O/${tag}_chain_bench.txt 2>&1
ls -la $O
