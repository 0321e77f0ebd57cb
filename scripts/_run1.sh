python -m pytest tests -m gpu -q -x > gpurun_out/r3_tests4.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3_tests4.log; tail -4 gpurun_out/r3_tests4.log
python scripts/gpu_ab.py TS2D_Q16=0 TS2D_Q16=1 --ops enc1.c1,enc2.c1,enc3.c1,enc4.c1,dec1.c1,dec2.c1,dec3.c1,dec4.c1 > gpurun_out/r3_ab_q16.txt 2>&1; cat gpurun_out/r3_ab_q16.txt
R=$(pwd); cd /tmp && export TMPDIR=/tmp
for q in 0 1; do
TS2D_Q16=$q timeout -k 10 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmc_q16_$q -o run -- python3 $R/scripts/gpu_ops_only.py split > $R/gpurun_out/pmc_q16_$q.log 2>&1
done
cd $R; ls gpurun_out/pmc_q16_0 | head
