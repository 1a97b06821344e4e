"""mean PMC counters per kernel from the passes of tools/pmc_run.sh: python tools/pmc_kernel.py <outdir> <kernel substring> [...]"""
import collections, csv, glob, json, sys
out, pats = sys.argv[1], sys.argv[2:]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        for p in pats:
            if p in k:
                agg[p][r['Counter_Name']].append(float(r['Counter_Value']))
                agg[p]['dur_us'].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
res = {}
for p, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    if 'FETCH_SIZE' in m:
        m['hbm_read_MB_corrected'] = 2 * m['FETCH_SIZE'] * 1024 / 1e6
    if 'WRITE_SIZE' in m:
        m['hbm_write_MB'] = m['WRITE_SIZE'] * 1024 / 1e6
    res[p] = m
print(json.dumps(res, indent=1))
