cd $GRAFT_REPO_ROOT
timeout -k 10 600 python bench.py --gpus 4 --one-device --steps 5 --warmup 2 --long-steps 32 > gpurun_out/rehearsal_4ranks.json 2> gpurun_out/rehearsal_4ranks.err; echo rc=$?
tail -c 1500 gpurun_out/rehearsal_4ranks.json; tail -5 gpurun_out/rehearsal_4ranks.err
timeout -k 10 300 python bench.py --force-collectives --steps 20 --no-cpu-baseline > gpurun_out/force_coll.json 2> gpurun_out/force_coll.err; echo rc=$?
python -c "
import json; d=json.load(open('gpurun_out/force_coll.json')); print(d['ms_per_step'], d['config']['collectives_ms_per_step'], d['config']['backend'], d['kernels_ms'])"
