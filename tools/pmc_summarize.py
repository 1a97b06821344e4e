"""Summarise the rocprofv3 PMC passes of tools/pmc_flash.sh into profiles/<name>.json (HBM traffic per launch of the
fused scoring kernels; FETCH_SIZE / WRITE_SIZE are KiB per dispatch, gfx950: FETCH_SIZE counts 64 B per 128-B request)."""
import collections, csv, glob, json, sys
out, dst, B, V, d, dtype = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'flash_ce' not in k:
            continue
        key = ('KIND_FWD' if ', 0>' in k else 'KIND_BWD') + '_grid%s' % r['Grid_Size']
        agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
        agg[key]['dur_us'].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
res = {}
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    e = dict(avg_duration_us_under_pmc=m.get('dur_us'), FETCH_SIZE_KiB=m.get('FETCH_SIZE'), WRITE_SIZE_KiB=m.get('WRITE_SIZE'))
    if 'FETCH_SIZE' in m and 'WRITE_SIZE' in m:
        e['hbm_read_bytes_corrected'] = 2 * m['FETCH_SIZE'] * 1024
        e['hbm_write_bytes'] = m['WRITE_SIZE'] * 1024
        e['traffic_bytes'] = e['hbm_read_bytes_corrected'] + e['hbm_write_bytes']
    for n in ('SQ_INSTS_VALU', 'SQ_INSTS_MFMA', 'SQ_INSTS_LDS', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'SQ_VALU_MFMA_BUSY_CYCLES',
              'SQ_BUSY_CU_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_ANY', 'SQ_ACTIVE_INST_ANY', 'GRBM_GUI_ACTIVE'):
        if n in m:
            e[n] = m[n]
    res[k] = e
# the merged backward launch is the largest KIND_BWD grid
bw = max((k for k in res if k.startswith('KIND_BWD')), key=lambda k: int(k.split('grid')[1]), default=None)
if bw:
    res['KIND_BWD'] = res[bw]
json.dump(dict(workload=dict(V=V, d=d, B=B, dtype=dtype),
               note='rocprofv3 --pmc <group> --kernel-trace, one counter group per pass (tools/pmc_flash.sh), python bench.py '
                    '--kernel-only; FETCH_SIZE/WRITE_SIZE KiB per dispatch; gfx950 FETCH correction: hbm_read = 2*FETCH_SIZE*1024 '
                    '(MI355X_MICROARCH.md, HBM).  KIND_BWD = the merged dE + d-sr launch (largest grid).',
               kernels=res), open(dst, 'w'), indent=1)
print(json.dumps(res.get('KIND_BWD'), indent=1))
