#!/bin/bash
# PMC passes over `bench.py --kernel-only` (fused scoring kernels alone); one counter group per rocprofv3 run.
# usage: tools/pmc_flash.sh <outdir> [bench args...]
out=$1; shift; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/p$i -- python /root/repo/bench.py --kernel-only "$@" > $out/p$i.log 2>&1 || echo "group $i ($grp) failed"
done
python - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'flash_ce' not in k and 'dsr_reduce' not in k and 'prepare' not in k:
            continue
        k = k.replace('(anonymous namespace)::', '').replace('void ', '')[:60] + ' grid=' + r['Grid_Size']
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
        agg[k]['dur_us'].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, d in sorted(agg.items()):
    print(k)
    for c, v in sorted(d.items()):
        print('   %-28s %14.1f  (n=%d)' % (c, sum(v) / len(v), len(v)))
PY
