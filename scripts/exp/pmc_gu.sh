set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for cfg in "64 212 fp8" "64 212 bf16" "32 8 fp8"; do
  tag=$(echo $cfg | tr ' ' '_')
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/gu_${tag}_p1 -- python3 scripts/exp/run_gu_once.py $cfg > gpurun_out/gu_${tag}_p1.log 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA --output-format csv -d gpurun_out/gu_${tag}_p2 -- python3 scripts/exp/run_gu_once.py $cfg > gpurun_out/gu_${tag}_p2.log 2>&1 || exit 2
  python3 scripts/pmc_kernel_mix.py gpurun_out/gu_${tag}_p1 linear_tiled > gpurun_out/gu_${tag}_p1.json || exit 3
  python3 scripts/pmc_kernel_mix.py gpurun_out/gu_${tag}_p2 linear_tiled > gpurun_out/gu_${tag}_p2.json || exit 4
  rm -rf gpurun_out/gu_${tag}_p1 gpurun_out/gu_${tag}_p2
done
echo done
