# round 2, GPU run 1: full -m gpu suite (incl. the full-size tests), flip rate, size sweep (the "before" numbers)
mkdir -p gpurun_out
free -g | head -2 > gpurun_out/host.txt; nproc >> gpurun_out/host.txt; cat /sys/fs/cgroup/cpu.max >> gpurun_out/host.txt 2>/dev/null
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/host.txt
tail -25 gpurun_out/pytest.log
timeout -k 10 600 python scripts/flip_rate.py --out gpurun_out/r02_flip_rate.txt > gpurun_out/flip.log 2>&1; echo "flip rc=$?" >> gpurun_out/host.txt
cat gpurun_out/r02_flip_rate.txt
