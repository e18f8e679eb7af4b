mkdir -p gpurun_out; L=gpurun_out/r04_fullsize.log; rm -f $L
(free -g; nproc; cat /sys/fs/cgroup/memory.max 2>/dev/null; cat /sys/fs/cgroup/cpu.max 2>/dev/null) 2>&1 | tee -a $L
for t in "tests/test_gpu_fullsize.py::test_playroom_calibration_lm_on_1M_events" "tests/test_gpu_fullsize.py::test_city_shape_lm_to_convergence_and_poisson" "tests/test_gpu_sharded.py::test_eight_ranks_at_full_shard_size"; do
  timeout -k 10 900 python -m pytest "$t" -x -q -s -m gpu > gpurun_out/r04_fs_one.log 2>&1; rc=$?
  echo "=== $t rc=$rc after ${SECONDS}s" | tee -a $L
  grep -E "passed|failed|Error|assert|Maximum resident|Elapsed|city shape:|playroom calibration:" gpurun_out/r04_fs_one.log | tee -a $L
  [ $rc -ne 0 ] && { tail -30 gpurun_out/r04_fs_one.log | tee -a $L; exit $rc; }
done
exit 0
