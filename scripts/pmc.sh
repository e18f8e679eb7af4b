# one rocprofv3 PMC pass over the bench; usage: PMC="A B C" TAG=name ARGS="bench args" bash scripts/pmc.sh
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG:-x}
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline $ARGS > $OUT/bench.log 2>&1 || { tail -5 $OUT/bench.log; exit 1; }
python3 - <<PY
import csv,glob,collections
f=glob.glob('$OUT/*/*_counter_collection.csv')[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'emba' in r['Kernel_Name']: agg[(r['Kernel_Name'].split('(')[0][-28:], r['Counter_Name'])].append(float(r['Counter_Value']))
for k in sorted(agg): print("%-30s %-28s %16.0f"%(k[0],k[1],sum(agg[k])/len(agg[k])))
PY
