#!/bin/bash
# kernel stats of tools/hgwin_check.py: tools/hgwin_prof.sh TAG [args] -> gpurun_out/TAG_hgwin_{log.txt,kernel_stats.txt}
tag=${1:-hgwin}; shift
cd /tmp && export TMPDIR=/tmp
out=/root/repo/gpurun_out
mkdir -p $out
rm -rf /tmp/prof_$tag
python /root/repo/tools/hgwin_check.py "$@" > $out/${tag}_hgwin_log.txt 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python /root/repo/tools/hgwin_check.py --iters 10 "$@" > /dev/null 2>&1
f=$(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1)
python - "$f" > $out/${tag}_hgwin_kernel_stats.txt <<PY
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:40]:
    print('%-90s calls %5s avg %9.1f us min %9.1f us' % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
PY
cat $out/${tag}_hgwin_log.txt | head -30
head -30 $out/${tag}_hgwin_kernel_stats.txt
