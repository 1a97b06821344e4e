"""per-kernel spread of durations over the replays of a trace: python tools/kernel_spread.py kernel_trace.csv
(min / median / max in us per launch position inside the step; kernels launched n times per step are split by position)"""
import csv, re, sys
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ends = [i for i, r in enumerate(rows) if 'adam_rows_kernel' in r['Kernel_Name']]
steps = [rows[a + 1:b + 1] for a, b in zip(ends[:-1], ends[1:])]
n = max(set(len(s) for s in steps), key=[len(s) for s in steps].count)
steps = [s for s in steps if len(s) == n][5:]            # replays of the captured step, warm-up dropped
print('%d replays of %d kernels' % (len(steps), n))
tot = np.array([sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in s) / 1e3 for s in steps])
print('busy per step: min %.1f median %.1f max %.1f' % (tot.min(), np.median(tot), tot.max()))
for i in range(n):
    d = np.array([(int(s[i]['End_Timestamp']) - int(s[i]['Start_Timestamp'])) / 1e3 for s in steps])
    name = re.sub(r'\(anonymous namespace\)::', '', steps[0][i]['Kernel_Name'])
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(.*$', '', name)[:60]
    print('%6.1f %6.1f %6.1f  %5.1f  %s' % (d.min(), np.median(d), d.max(), d.max() - d.min(), name))
