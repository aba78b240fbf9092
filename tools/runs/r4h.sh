mkdir -p gpurun_out/r4h
L="--lib $PWD/squigulator_amd/csrc/libsqg_hip_dev.so --pipeline-seconds 0 --e2e-seconds 0"
for abl in 2 1 0; do
echo "== SQG_PHC_ABL=$abl"
SQG_PHC_ABL=$abl bash tools/pmc_quick.sh "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" $L | grep "hand\|events<1"
SQG_PHC_ABL=$abl bash tools/pmc_quick.sh "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU" $L | grep "hand\|events<1"
done > gpurun_out/r4h/pmc.log 2>&1
cat gpurun_out/r4h/pmc.log
