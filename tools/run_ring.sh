mkdir -p gpurun_out
for m in side kernel main; do
SREC_STAGE_MODE=$m python bench.py --e2e-loader ring --e2e-probe > gpurun_out/r03_stage_$m.json 2> gpurun_out/r03_stage_$m.err || tail -20 gpurun_out/r03_stage_$m.err
done
python - <<'PY'
import json
for m in ('side','kernel','main'):
    j=json.loads(open('gpurun_out/r03_stage_%s.json'%m).read().strip().splitlines()[-1])
    e=j['end_to_end']; print(m, round(j['ms_per_step'],4), round(e['ms_per_step'],4), e['final_loss'], e['loader_waits_ms_per_step'], e['probe'])
PY
