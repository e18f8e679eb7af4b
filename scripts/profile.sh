# rocprofv3 passes for the judged numbers: kernel-trace stats, then FETCH_SIZE and WRITE_SIZE in separate PMC passes
# (MI355X_MICROARCH.md: the two do not fit one pass).  Outputs under gpurun_out/prof_$TAG; copy summaries to profiles/.
TAG=${TAG:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_trace.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_write.log 2>&1 || exit 1
find $OUT -name "*.csv" | head -20
