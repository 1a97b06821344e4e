"""one training step as a launch sequence: python tools/step_trace.py kernel_trace.csv  (the last adam_rows-to-adam_rows span)"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ends = [i for i, r in enumerate(rows) if 'adam_rows_kernel' in r['Kernel_Name']]
if len(ends) < 3:
    sys.exit('fewer than 3 steps in the trace')
lo, hi = ends[-3] + 1, ends[-2] + 1          # a replayed step in the middle of the timed region
t0 = int(rows[lo]['Start_Timestamp'])
prev_end = None
print('%6s %8s %7s %6s  %s' % ('start', 'dur_us', 'gap_us', 'grid', 'kernel'))
tot = 0.0
for r in rows[lo:hi]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(.*$', '', name)[:70]
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    grid = r.get('Grid_Size', r.get('Grid_Size_X', ''))
    print('%6.1f %8.1f %7.1f %6s  %s' % ((s - t0) / 1e3, (e - s) / 1e3, gap, grid, name))
    tot += (e - s) / 1e3
    prev_end = e
print('kernels %d, busy %.1f us, span %.1f us' % (hi - lo, tot, (int(rows[hi - 1]['End_Timestamp']) - t0) / 1e3))
# large gaps of the last few steps (a gap inside a replayed graph = a node that is not a kernel: memset / memcpy)
print('gaps > 3 us per step (start offset, gap, next kernel):')
for a, b in list(zip(ends[:-1], ends[1:]))[-6:]:
    t0s = int(rows[a + 1]['Start_Timestamp'])
    out = []
    for i in range(a + 2, b + 1):
        gap = (int(rows[i]['Start_Timestamp']) - int(rows[i - 1]['End_Timestamp'])) / 1e3
        if gap > 3.0:
            out.append('%.0f:%.1f:%s' % ((int(rows[i]['Start_Timestamp']) - t0s) / 1e3, gap, re.sub(r'\(.*$', '', rows[i]['Kernel_Name'])[-28:]))
    print('  ', ' | '.join(out))
