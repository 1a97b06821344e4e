"""durations of one kernel over the launches of a trace, in launch order: python tools/kernel_series.py kernel_trace.csv NAME_SUBSTRING
(and the step period = distance between consecutive adam_rows ends)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
pat = sys.argv[2]
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if pat in r['Kernel_Name']]
print(pat, len(d), 'launches:', ' '.join('%.1f' % x for x in d))
e = [int(r['End_Timestamp']) for r in rows if 'adam_rows_kernel' in r['Kernel_Name']]
print('step period us:', ' '.join('%.0f' % ((b - a) / 1e3) for a, b in zip(e[:-1], e[1:])))
