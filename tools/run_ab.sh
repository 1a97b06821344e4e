mkdir -p gpurun_out
for m in 0 1 0 1; do
SREC_FWD_TILED=$m python bench.py --no-end-to-end 2> gpurun_out/ab_$m.err | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tiled=$m', round(j['ms_per_step'],4), round(j['value']))"
done
python tools/fwd_wres_bench.py
