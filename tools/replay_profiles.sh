#!/bin/bash
# transport-free prediction of the 2 / 4 / 8-GPU points on ONE GPU (bench.py --replay-rank R --of W): for W in 2 4 8 and both
# scaling modes the JSON line (kernel ms of the rank's captured step, bytes per exchange) and a kernel trace of ONE replayed step
#   tools/replay_profiles.sh rNN [ranks...]   ->   gpurun_out/rNN_rank{W}_{weak,strong}_{bench.json,step_trace.txt,breakdown.txt}
tag=${1:-r06}; shift
worlds=${@:-2 4 8}
out=/root/repo/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for W in $worlds; do
  R=$((W - 1))
  for mode in weak strong; do
    extra=""; [ $mode = strong ] && extra="--global-batch 512"
    name=${tag}_rank${W}_${mode}
    timeout -k 10 900 python /root/repo/bench.py --replay-rank $R --of $W $extra --steps 30 --warmup 5 > $out/${name}_bench.json 2> $out/${name}_bench.err
    rm -rf /tmp/prof_$name
    timeout -k 10 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$name -- python /root/repo/bench.py --replay-rank $R --of $W $extra --steps 20 --warmup 5 --repeats 1 > $out/${name}_prof.log 2>&1
    # (the recording job's rank processes are traced too: the replaying parent's trace is the one with the most kernel rows)
    t=$(for f in $(find /tmp/prof_$name -name '*kernel_trace.csv'); do echo "$(wc -l < $f) $f"; done | sort -n | tail -1 | cut -d' ' -f2)
    [ -n "$t" ] && python /root/repo/tools/step_trace.py "$t" > $out/${name}_step_trace.txt
    [ -n "$t" ] && python /root/repo/tools/step_breakdown.py "$t" > $out/${name}_breakdown.txt
    echo "$name: $(cut -c1-300 $out/${name}_bench.json)"
  done
done
