mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "gemm_f32_group or readout_head" 2>&1 | tail -4
for m in 1 0 1 0; do
SREC_GEMM_FUSED_REDUCE=$m python bench.py --no-end-to-end 2> gpurun_out/ab_$m.err | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused_reduce=$m', round(j['ms_per_step'],4), round(j['value']), j.get('graph_nodes'))"
done
