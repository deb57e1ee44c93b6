#!/usr/bin/env python3
"""Mean duration of the k-th kernel of a Jacobian step from a rocprofv3 kernel trace (kernels back to back on one stream:
PJ_RBLK_SPLIT=0).  usage: kernel_seq.py <kernel_trace.csv> <kernels per step>"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'k_rblk' in r['Kernel_Name'] or 'k_pre' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
per = int(sys.argv[2])
n = len(rows) // per
rows = rows[len(rows) - n * per:]
out = []
for k in range(per):
    d = [(int(rows[q * per + k]['End_Timestamp']) - int(rows[q * per + k]['Start_Timestamp'])) / 1e6 for q in range(n)]
    out.append((rows[k]['Kernel_Name'].split('::')[-1][:6], round(sum(d) / len(d), 3)))
print(n, 'steps:', out, 'sum', round(sum(x[1] for x in out), 3))
