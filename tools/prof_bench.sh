#!/bin/bash
# kernel-stats profile of the bench step: tools/prof_bench.sh TAG [extra bench args] ->  gpurun_out/TAG_{bench.json,kernel_stats.csv}
# (one timed region, no fp32 side run: every kernel row of the stats belongs to the bf16 step; 27 step executions =
#  2 eager capture warm-ups + 5 warm-up replays + 20 timed replays)
tag=${1:-prof}; shift
cd /tmp && export TMPDIR=/tmp
out=/root/repo/gpurun_out
mkdir -p $out
rm -rf /tmp/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python /root/repo/bench.py --steps 20 --warmup 5 --step-only "$@" > $out/${tag}_bench.log 2>&1
grep '^{' $out/${tag}_bench.log | tail -1 > $out/${tag}_bench.json
f=$(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1)
cp "$f" $out/${tag}_kernel_stats.csv
t=$(find /tmp/prof_$tag -name '*kernel_trace.csv' | head -1)
[ -n "$t" ] && python /root/repo/tools/step_trace.py "$t" > $out/${tag}_step_trace.txt
[ -n "$t" ] && python /root/repo/tools/step_breakdown.py "$t" > $out/${tag}_breakdown.txt
head -70 $out/${tag}_breakdown.txt
