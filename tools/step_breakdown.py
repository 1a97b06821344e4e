"""per-step kernel time by group, from a rocprofv3 kernel_trace.csv: python tools/step_breakdown.py kernel_trace.csv
Groups the kernels of ONE replayed step in the middle of the timed region - the same adam_rows-to-adam_rows span
tools/step_trace.py lists - so the total equals the step trace's busy time and nothing of the set-up (eager warm-up steps with
their fill / copy kernels, the dominant-kernel timing loop) is booked to the step."""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ends = [i for i, r in enumerate(rows) if 'adam_rows_kernel' in r['Kernel_Name']]
if len(ends) < 3:
    sys.exit('fewer than 3 steps in the trace')
lo, hi = ends[-3] + 1, ends[-2] + 1


def short(k):
    k = re.sub(r'\(anonymous namespace\)::', '', k)
    k = re.sub(r'^void ', '', k)
    k = re.sub(r'\(.*$', '', k)
    k = k.replace('at::native::', 'aten::')
    return k[:70]


GROUPS = [('prologue (weight copies + intake)', r'step_prep'),
          ('scoring', r'flash_ce|ce_reduce|ce_mean|dsr_reduce|bf16_prepare|rownorm_project|row_invnorm|renorm_rows_bf16'),
          ('adam', r'adam'),
          ('readout head', r'head_fwd|head_bwd|head_wfrag|gemm_f32_group|splitk_reduce_group|seg_attn|cat_cols|normalize_fwd|normalize_bwd'),
          ('bf16 GEMMs (GAT + GRU)', r'gemm_group|gemm16|rows_bf16|weights_bf16|sum_slabs'),
          ('gat_graph', r'hg_'),
          ('gru', r'gru_|gram_|gemm_bf16_tn|splitk|gemm_bf16_nt|gemm_f32'),
          ('rows', r'normalize|gather_rows|scatter_add|col_sum|renorm|mask_scale|permute|pick|copy_words'),
          ('aten / runtime', r'at::native|aten|rocclr')]
per = collections.OrderedDict()
tot = 0.0
for r in rows[lo:hi]:
    t = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    g = next((g for g, pat in GROUPS if re.search(pat, r['Kernel_Name'])), 'misc')
    d = per.setdefault(g, collections.OrderedDict())
    k = short(r['Kernel_Name'])
    c, s = d.get(k, (0, 0.0))
    d[k] = (c + 1, s + t)
    tot += t
span = (int(rows[hi - 1]['End_Timestamp']) - int(rows[lo]['Start_Timestamp'])) / 1e3
print('one replayed step: %d kernels, busy %.1f us, span %.1f us' % (hi - lo, tot, span))
for g, d in sorted(per.items(), key=lambda kv: -sum(s for _, s in kv[1].values())):
    print('== %-24s %7.1f us  %3d launches' % (g, sum(s for _, s in d.values()), sum(c for c, _ in d.values())))
    for name, (c, s) in sorted(d.items(), key=lambda x: -x[1][1]):
        print('     %-72s x%-3d %7.1f' % (name, c, s))
