"""Summarise tools/pmc_run.sh passes into one json: per kernel (name substring filter) the mean of every counter, the mean
duration under PMC and the HBM traffic per launch (gfx950: FETCH_SIZE counts 64 B per 128-B request -> read bytes =
2 * FETCH_SIZE KiB * 1024; MI355X_MICROARCH.md, HBM).
usage: python tools/pmc_json.py <outdir> <dst.json> <note> <filter> [<filter> ...]"""
import collections, csv, glob, json, re, sys
out, dst, note, filters = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4:]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if not any(x in k for x in filters):
            continue
        k = re.sub(r'\(anonymous namespace\)::', '', k)
        k = re.sub(r'^void ', '', k)
        k = re.sub(r'\(.*$', '', k) + ' grid=' + r['Grid_Size']
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
        agg[k]['dur_us'].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
res = {}
for k, c in sorted(agg.items()):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    e = dict(avg_duration_us_under_pmc=m.pop('dur_us', None), launches_sampled=len(c['dur_us']))
    if 'FETCH_SIZE' in m and 'WRITE_SIZE' in m:
        e['hbm_read_bytes_corrected'] = 2 * m['FETCH_SIZE'] * 1024
        e['hbm_write_bytes'] = m['WRITE_SIZE'] * 1024
        e['traffic_bytes'] = e['hbm_read_bytes_corrected'] + e['hbm_write_bytes']
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in m and 'GRBM_GUI_ACTIVE' in m:
        # MFMA pipe busy fraction: busy cycles summed over the 1024 SIMDs / (kernel cycles * 1024)
        e['mfma_busy_frac'] = m['SQ_VALU_MFMA_BUSY_CYCLES'] / (m['GRBM_GUI_ACTIVE'] * 1024 / 8) if m['GRBM_GUI_ACTIVE'] else None
    e.update({('FETCH_SIZE_KiB' if n == 'FETCH_SIZE' else 'WRITE_SIZE_KiB' if n == 'WRITE_SIZE' else n): v for n, v in m.items()})
    res[k] = e
json.dump(dict(note=note, kernels=res), open(dst, 'w'), indent=1)
for k, e in res.items():
    print(k, {x: (round(v, 1) if isinstance(v, float) else v) for x, v in e.items() if x in ('avg_duration_us_under_pmc', 'traffic_bytes', 'mfma_busy_frac', 'SQ_LDS_BANK_CONFLICT')})
