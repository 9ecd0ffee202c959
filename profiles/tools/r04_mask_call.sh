mkdir -p gpurun_out/mask
timeout 600 python profiles/tools/ab_run.py --workloads mixed text records samples16 --reps 3 --out gpurun_out/mask/ab.json > gpurun_out/mask/ab.txt 2>&1
cat gpurun_out/mask/ab.txt
PMC_SET1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY" bash profiles/tools/pmc_variants.sh mask_pmc base $VARIANTS 2>&1 | tail -12
