#!/bin/bash
# rocprofv3 counter passes over the TRAINING step (tools/train_step_bench.py), one --pmc set per run, kernel-trace only
# usage on the GPU box:  bash tools/pmc_passes_train.sh  -> gpurun_out/pmct/pmc_<SET>/ ; summarise with
#   python tools/pmc_summary.py gpurun_out/pmct profiles/r02_train_wgrad_pmc.json conv_wgrad_img_kernel
R=$PWD
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmct
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmct/pmc_$tag -o pmc -- python3 $R/tools/train_step_bench.py --batch 16 --steps 1 --warmup 1 > $R/gpurun_out/pmct/pmc_$tag.log 2>&1 || echo "pmc $tag failed"
done
ls $R/gpurun_out/pmct
