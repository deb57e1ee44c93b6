#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection csv: per kernel name, mean of each counter."""
import csv, sys, collections, glob, json, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r'\bk_\w+(<\d+>)?', r['Kernel_Name'])
            k = m.group(0) if m else r['Kernel_Name'][:24]
            if m and float(r.get('Grid_Size', 0) or 0) < 100000: k += ':small'
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
out = {k: {c: dict(n=len(v), mean=sum(v) / len(v)) for c, v in cs.items()} for k, cs in acc.items()}
print(json.dumps(out, indent=1))
