#!/bin/bash
# A / B of an environment switch over the replays of the bench step: tools/ab_env.sh VAR "v1 v2 ..." 'grep pattern'
cd /tmp && export TMPDIR=/tmp
for v in $2; do
  rm -rf /tmp/ks; env $1=$v timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ks -- python /root/repo/bench.py --steps 60 --warmup 5 --step-only > /tmp/o.txt 2>&1
  t=$(find /tmp/ks -name "*kernel_trace.csv" | head -1); echo "$1=$v"; python /root/repo/tools/kernel_spread.py $t | grep -E "busy|$3"
done
