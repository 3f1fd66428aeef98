#!/bin/bash
# rocprofv3 counter passes for profiles/ (each --pmc set in its own run, kernel-trace only -- no other trace domain)
# usage on the GPU box:  bash tools/pmc_passes.sh [suffix]  -> gpurun_out/pmc<suffix>_<SET>/ ; summarise with tools/pmc_summary.py
# (environment, e.g. CDDPM_BENCH_ACCUM_SWITCH=0 for the opt-in two-level accumulation plan, is inherited by bench.py)
R=$PWD
SUF=$1
cd /tmp && export TMPDIR=/tmp
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc${SUF}_$tag -o pmc -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --no-profile --no-alt > $R/gpurun_out/pmc${SUF}_$tag.log 2>&1 || echo "pmc $tag failed"
done
ls $R/gpurun_out/ | grep pmc${SUF}_
