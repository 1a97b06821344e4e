mkdir -p gpurun_out
for m in 0 1 0 1; do
SREC_EARLY_ADAM=$m SREC_DEBUG_CAPTURE=1 python bench.py --no-end-to-end 2> gpurun_out/early_$m.err | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('early=$m', j['ms_per_step'], j['value'], j.get('kernels_per_step'), j.get('launch_mode'))"
grep -i "error\|fail\|Traceback" gpurun_out/early_$m.err | head -5
done
