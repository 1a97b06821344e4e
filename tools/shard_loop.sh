#!/bin/bash
# repeat the 1-rank sharded bench N times, count failures
n=${1:-6}; fail=0
for i in $(seq 1 $n); do
  SREC_FORCE_COLLECTIVES=1 timeout 120 python bench.py --shard --step-only --steps 20 --warmup 5 > /tmp/sl_$i.out 2> /tmp/sl_$i.err
  rc=$?
  if [ $rc -ne 0 ]; then fail=$((fail+1)); echo "run $i rc=$rc: $(grep -m1 -E 'HIP error|Error' /tmp/sl_$i.err | cut -c1-200)"; else echo "run $i ok $(tail -1 /tmp/sl_$i.out | cut -c1-80)"; fi
done
echo "failures: $fail / $n"
