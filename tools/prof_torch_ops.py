"""List the torch (non-library) GPU ops of one eager training step with their input shapes."""
import importlib, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from torch.profiler import profile, ProfilerActivity

def main():
    sys.argv = ['bench.py', '--no-graph', '--no-cpu-baseline', '--steps', '2', '--warmup', '2'] + sys.argv[1:]
    orig = bench.time_dominant_kernel
    bench.time_dominant_kernel = lambda *a, **k: {'bf16': False, 'dE': 1.0, 'dsr': 1.0, 'fwd': 1.0}
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
        bench.main()
    rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith('aten::')]
    rows.sort(key=lambda e: -e.count)
    for e in rows[:60]:
        print('%-28s n=%4d  %s' % (e.key, e.count, str(e.input_shapes)[:150]))

main()
