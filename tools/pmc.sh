#!/bin/bash
# rocprofv3 PMC passes over bench.py (separate passes: TCC slots do not fit FETCH_SIZE and WRITE_SIZE together)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --sub-steps 0 > $R/gpurun_out/pmc_$tag.log 2>&1 || echo "pass $tag failed"
done
ls $R/gpurun_out/pmc_*/*/ | head -30
