# counters of the tiled kernel after the per-event pose: traffic + where wave time goes
set -e
mkdir -p gpurun_out
A="--events-per-gpu 100000000 --knots 256 --pano-h 2048"
TAG=r02c_100M ARGS="$A" STEPS=4 bash scripts/profile.sh > gpurun_out/prof_100M.log 2>&1 || { tail gpurun_out/prof_100M.log; exit 1; }
PMC="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES" TAG=sq1_100M ARGS="$A" bash scripts/pmc.sh > gpurun_out/pmc_sq1.txt 2>&1 || tail gpurun_out/pmc_sq1.txt
PMC="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU" TAG=sq2_100M ARGS="$A" bash scripts/pmc.sh > gpurun_out/pmc_sq2.txt 2>&1 || tail gpurun_out/pmc_sq2.txt
PMC="TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" TAG=tcc_100M ARGS="$A" bash scripts/pmc.sh > gpurun_out/pmc_tcc.txt 2>&1 || tail gpurun_out/pmc_tcc.txt
grep -h "warp_tiled\|gram_kernel" gpurun_out/pmc_sq1.txt gpurun_out/pmc_sq2.txt gpurun_out/pmc_tcc.txt
