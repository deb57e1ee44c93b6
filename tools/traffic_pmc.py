#!/usr/bin/env python3
"""HBM traffic of one Jacobian step from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE in separate
runs of tools/one_step.py).  Sums over all kernels of the step (the pj_rows path has several);
gfx950: FETCH_SIZE counts half (MI355X_MICROARCH.md HBM section), both counters are in KiB... as
reported by rocprofv3 (units of 1 KB).
usage: traffic_pmc.py <fetch_dir> <write_dir> <steps> <states> <bytes_per_state> <label> [library file name] > profiles/traffic_X.json
(the library file name carries the digest of kernel sources + build options: bench.py compares it with the attached library)"""
import csv, glob, json, re, sys
fd, wd, steps, states, bps, label = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
library = sys.argv[7] if len(sys.argv) > 7 and sys.argv[7] not in ('', '-') else None


def total(d, counter):
    tot, per = 0.0, {}
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r'\bk_\w+', r['Kernel_Name'])
            if r['Counter_Name'] == counter and m:
                tot += float(r['Counter_Value'])
                per[m.group(0)] = per.get(m.group(0), 0.0) + float(r['Counter_Value'])
    return tot, per


f, fper = total(fd, 'FETCH_SIZE')
w, wper = total(wd, 'WRITE_SIZE')
hbm = (2.0 * f + w) * 1024.0 / steps
alg = states * bps
print(json.dumps({
    'workload': label, 'library': library, 'states_per_launch': states, 'steps_profiled': steps,
    'FETCH_SIZE_KB_per_step': f / steps, 'WRITE_SIZE_KB_per_step': w / steps,
    'per_kernel_KB_per_step': {k: {'FETCH_SIZE': fper.get(k, 0) / steps, 'WRITE_SIZE': wper.get(k, 0) / steps}
                               for k in sorted(set(fper) | set(wper))},
    'hbm_bytes_per_launch': hbm, 'algorithmic_bytes_per_launch': alg, 'ratio': hbm / alg,
    'note': 'hbm = (2*FETCH_SIZE + WRITE_SIZE) * 1024 summed over every kernel of a Jacobian step '
            '(gfx950 FETCH_SIZE half-count correction); separate --pmc passes; tools/one_step.py'}, indent=1))
