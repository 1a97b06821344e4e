#!/bin/bash
# kernel sequence of the LAST graph replay of the default bench -> gpurun_out/seq.txt   (extra bench args: "$@")
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trace_seq
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_seq -- python /root/repo/bench.py --steps 3 --warmup 2 --no-cpu-baseline "$@" > /tmp/trace_seq.log 2>&1
f=$(find /tmp/trace_seq -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# last occurrence of adam_rows marks the end of a step; take the window between the last two
idx = [i for i, n in enumerate(names) if 'adam_rows' in n]
# steps inside the timed region: pick the last replayed step before time_dominant_kernel's loops
import re
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    return n[:90]
# choose the 3rd from last adam_rows .. 2nd from last as the window
cands = [(idx[i], idx[i + 1]) for i in range(len(idx) - 1)]
best = [c for c in cands if 100 < c[1] - c[0] < 400]
a, b = best[-1]
t0 = int(rows[a + 1]['Start_Timestamp'])
out = open('/root/repo/gpurun_out/seq.txt', 'w')
prev_end = None
for r in rows[a + 1:b + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = 0 if prev_end is None else s - prev_end
    out.write('%8.1f  dur %7.1f  gap %6.1f  %s\n' % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, short(r['Kernel_Name'])))
    prev_end = e
out.write('kernels %d, span %.1f us\n' % (b - a, (int(rows[b]['End_Timestamp']) - t0) / 1e3))
PY
