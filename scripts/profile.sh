# rocprofv3 passes for the judged numbers: kernel-trace stats, then FETCH_SIZE, WRITE_SIZE and the memory-side atomic requests in
# separate PMC passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass).  Outputs under gpurun_out/prof_$TAG;
# scripts/summarize_profiles.py copies the summaries to profiles/.
#   TAG=r02a ARGS="--events-per-gpu 10000000 --knots 97" STEPS=10 bash scripts/profile.sh
TAG=${TAG:-r02}
ARGS=${ARGS:-}
STEPS=${STEPS:-20}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps $STEPS --warmup 3 --no-cpu-baseline $ARGS > $OUT/bench_trace.log 2>&1 || { tail -5 $OUT/bench_trace.log; exit 1; }
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline $ARGS > $OUT/bench_fetch.log 2>&1 || { tail -5 $OUT/bench_fetch.log; exit 1; }
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline $ARGS > $OUT/bench_write.log 2>&1 || { tail -5 $OUT/bench_write.log; exit 1; }
timeout -k 10 400 rocprofv3 --pmc TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $OUT/atomic -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline $ARGS > $OUT/bench_atomic.log 2>&1 || { tail -5 $OUT/bench_atomic.log; echo "(atomic pass failed: continuing)"; }
cd $GRAFT_REPO_ROOT && python3 scripts/summarize_profiles.py $TAG
