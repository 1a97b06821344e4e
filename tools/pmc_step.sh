#!/bin/bash
# PMC passes over a short default bench (whole training step); one counter group per rocprofv3 run.
# usage: tools/pmc_step.sh "<kernel-name regex>"  ->  gpurun_out/pmc_step.txt
# NOTE: the TCC_* / TA_* / FETCH_SIZE groups abort rocprofv3 (signal 6) when the kernels run inside a replayed hipGraph on
# this ROCm build: collect those on eager launches (bench.py --no-graph, or tools/pmc_flash.sh for the scoring kernels).
pat=${1:-gemm_group}
cd /tmp && export TMPDIR=/tmp
out=/tmp/pmc_step
rm -rf $out; mkdir -p $out
groups=(
 "SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_BUSY_sum"
 "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAVES"
 "GRBM_GUI_ACTIVE FETCH_SIZE WRITE_SIZE"
)
i=0
for grp in "${groups[@]}"; do
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/p$i -- python /root/repo/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $out/p$i.log 2>&1 || echo "group $i failed: $(tail -2 $out/p$i.log)"
  i=$((i+1))
done
python - "$pat" <<'PY' > /root/repo/gpurun_out/pmc_step.txt
import csv, glob, re, sys, collections
pat = re.compile(sys.argv[1])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pmc_step/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if pat.search(n):
            n = re.sub(r'\(anonymous namespace\)::', '', n)[:90]
            acc[n][r['Counter_Name']].append(float(r['Counter_Value']))
for n, cs in acc.items():
    print(n)
    for c, v in sorted(cs.items()):
        print('   %-40s n=%3d  mean %.4g' % (c, len(v), sum(v) / len(v)))
PY
cat /root/repo/gpurun_out/pmc_step.txt
