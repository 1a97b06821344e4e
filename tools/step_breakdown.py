"""per-step kernel time from a rocprofv3 kernel_stats.csv: python tools/step_breakdown.py stats.csv N_STEP_EXECUTIONS
Kernels whose call count is a multiple of the step count are attributed to the step; the rest (timing loops of
bench.time_dominant_kernel, set-up) are listed separately."""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 27
def short(k):
    k = re.sub(r'\(anonymous namespace\)::', '', k)
    k = re.sub(r'^void ', '', k)
    k = re.sub(r'\(.*$', '', k)
    k = k.replace('at::native::', 'aten::')
    return k[:70]
GROUPS = [('scoring', r'flash_ce|ce_reduce|ce_mean|dsr_reduce|bf16_prepare|rownorm_project|row_invnorm|renorm_rows_bf16'),
          ('adam', r'adam'),
          ('readout head', r'gemm_f32_group|splitk_reduce_group|seg_attn|cat_cols'),
          ('bf16 GEMMs (GAT + GRU)', r'gemm_group|gemm16|rows_bf16|weights_bf16|sum_slabs'),
          ('gat_graph', r'hg_'),
          ('gru steps', r'gru_|gram_|gemm_bf16_tn|splitk|gemm_bf16_nt|gemm_f32'),
          ('rows', r'normalize|gather_rows|scatter_add|col_sum|renorm|mask_scale|permute|pick'),
          ('aten', r'at::native|aten|rocclr')]
per, tot, other = collections.OrderedDict(), 0.0, []
for r in rows:
    calls, t = int(r['Calls']), float(r['TotalDurationNs']) / 1e3
    if calls % n:
        # mixed: per-step part = floor(calls/n) * n calls at the average duration
        k = calls // n
        if k == 0:
            other.append((short(r['Name']), calls, t))
            continue
        t_step = k * float(r['AverageNs']) / 1e3
        other.append((short(r['Name']) + ' [non-step part]', calls - k * n, t - t_step * n))
    else:
        k, t_step = calls // n, t / n
    g = next((g for g, pat in GROUPS if re.search(pat, r['Name'])), 'misc')
    per.setdefault(g, []).append((short(r['Name']), k, t_step))
    tot += t_step
print('step total %.1f us, %d launches' % (tot, sum(k for v in per.values() for _, k, _ in v)))
for g, v in per.items():
    print('== %-14s %7.1f us  %3d launches' % (g, sum(t for _, _, t in v), sum(k for _, k, _ in v)))
    for name, k, t in sorted(v, key=lambda x: -x[2]):
        print('     %-72s x%-3d %7.1f' % (name, k, t))
if other:
    print('-- not per step:', ', '.join('%s x%d %.0fus' % o for o in other[:12]))
