# round 4, experiment 5: anatomy of the small kernels by ablation (diag build), rocprof kernel durations
mkdir -p gpurun_out && rm -f gpurun_out/r04_exp5.log
L=gpurun_out/r04_exp5.log
export EMBA_LIB=$PWD/build_variants/diag.so
for a in 0 2048 4096 6144 8192 14336 16384; do
  echo "=== EMBA_ABLATE=$a" | tee -a $L
  EMBA_ABLATE=$a TAG=r04e_$a STEPS=100 bash scripts/quick_trace.sh 2>&1 | grep -E "active_write|post_warp_a|prep_pose|gram_compact|warp_residual" | tee -a $L
done
