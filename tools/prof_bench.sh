#!/bin/bash
# kernel-stats profile of the default bench: tools/prof_bench.sh TAG  ->  gpurun_out/TAG_{bench.json,kernel_stats.csv}
tag=${1:-prof}
cd /tmp && export TMPDIR=/tmp
out=/root/repo/gpurun_out
mkdir -p $out
rm -rf /tmp/prof_$tag
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python /root/repo/bench.py --steps 20 --warmup 5 > $out/${tag}_bench.log 2>&1
tail -1 $out/${tag}_bench.log > $out/${tag}_bench.json
f=$(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1)
cp "$f" $out/${tag}_kernel_stats.csv
head -40 $out/${tag}_kernel_stats.csv | cut -c1-150
