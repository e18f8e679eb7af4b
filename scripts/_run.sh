cd $GRAFT_REPO_ROOT
timeout -k 5 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "resident_step or ep_in_reference or irls or small_config or state_parity" > gpurun_out/t6.log 2>&1; tail -3 gpurun_out/t6.log
for i in 1 2; do
timeout -k 5 200 python bench.py --steps 20 --no-cpu-baseline > gpurun_out/bench_r05_e.json 2> gpurun_out/bench_r05_e.err
python -c "
import json; d=json.load(open('gpurun_out/bench_r05_e.json')); k=d['kernels_ms']; print(round(d['ms_per_step']*1e3,1), round(d['ms_per_step_long']*1e3,1), round(d['config']['no_ep_ms_per_step']*1e3,1), 'warp', round(d['roofline']['kernel_ms']*1e3,1), 'prep %.1f warp %.1f A %.1f gram %.1f'%(k['prep_pose_texel']*1e3,k['warp']*1e3,k['post_warp_a']*1e3,k['gram']*1e3), d['device']['sysfs']['sclk_mhz'])"
done
timeout -k 5 200 python bench.py --steps 20 --shard-of 8 --shard-rank 3 --no-cpu-baseline > gpurun_out/bench_r05_shard.json 2> gpurun_out/bench_r05_shard.err
python -c "
import json; d=json.load(open('gpurun_out/bench_r05_shard.json')); k=d['kernels_ms']; print('shard', round(d['ms_per_step']*1e3,1), round(d['ms_per_step_long']*1e3,1), round(d['config']['no_ep_ms_per_step']*1e3,1), 'prep %.1f warp %.1f A %.1f gram %.1f'%(k['prep_pose_texel']*1e3,k['warp']*1e3,k['post_warp_a']*1e3,k['gram']*1e3))"
