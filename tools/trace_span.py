#!/usr/bin/env python3
"""Step time of the row-block kernels from a rocprofv3 kernel trace (*_kernel_trace.csv).

With a batch running as two parts on two streams the kernels of a step overlap, so the sum of the
per-kernel average durations of *_kernel_stats.csv is no longer the step time.  This takes the
full-size launches (grid at least 40 % of the largest k_rblk grid: both parts, not the small
validation batch), and reports the time the device spent inside them (union of their intervals)
per step, next to the plain sum of durations.
usage: trace_span.py <kernel_trace.csv> <parts per step> [label]"""
import csv, json, sys


def main():
    path, parts = sys.argv[1], int(sys.argv[2])
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r['Kernel_Name']
            if 'k_rblk' in name or 'k_pre' in name:
                rows.append(('k_pre' if 'k_pre' in name else 'k_rblk', int(r['Start_Timestamp']), int(r['End_Timestamp']),
                             int(r['Grid_Size_X']) if 'Grid_Size_X' in r else int(r['Grid_Size'])))
    gmax = max(g for k, _, _, g in rows if k == 'k_rblk')
    rows = [r for r in rows if r[3] >= 0.4 * gmax]
    npre = sum(1 for r in rows if r[0] == 'k_pre')
    steps = npre / parts
    iv = sorted((a, b) for _, a, b, _ in rows)
    union, cur_a, cur_b = 0, iv[0][0], iv[0][1]
    for a, b in iv[1:]:
        if a > cur_b:
            union += cur_b - cur_a
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    union += cur_b - cur_a
    out = dict(label=sys.argv[3] if len(sys.argv) > 3 else path, parts_per_step=parts, steps=steps,
               launches=dict(k_pre=npre, k_rblk=len(rows) - npre),
               device_time_in_step_kernels_ms_per_step=union / steps / 1e6,
               sum_of_kernel_durations_ms_per_step=sum(b - a for _, a, b, _ in rows) / steps / 1e6,
               avg_ms=dict(k_pre=sum(b - a for k, a, b, _ in rows if k == 'k_pre') / max(npre, 1) / 1e6,
                           k_rblk=sum(b - a for k, a, b, _ in rows if k == 'k_rblk') / max(len(rows) - npre, 1) / 1e6),
               note='union of the kernel intervals / steps: what bench.py times with HIP events on the caller\'s stream')
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
