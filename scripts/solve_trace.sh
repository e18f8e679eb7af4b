# kernel trace of scripts/solve_timing.py -> per-kernel breakdown of one device Schur solve per configuration
#   TAG=r03c SOLVE_CONFIGS=1,3 bash scripts/solve_trace.sh
TAG=${TAG:-s}
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_solve_$TAG
rm -rf $OUT && mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/scripts/solve_timing.py > $OUT/timing.log 2>&1 || { tail -5 $OUT/timing.log; exit 1; }
cat $OUT/timing.log | grep "^N="
python3 $GRAFT_REPO_ROOT/scripts/solve_breakdown.py $OUT | tee $OUT/breakdown.txt
find $OUT -name "*_kernel_trace.csv" -size +20M -delete
