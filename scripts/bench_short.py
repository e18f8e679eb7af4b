import subprocess, sys, os
# (helper for trace_cmd.sh: bench.py with few steps, no CPU baseline)
sys.argv = ["bench.py", "--steps", "20", "--warmup", "3", "--no-cpu-baseline"]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import runpy
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
