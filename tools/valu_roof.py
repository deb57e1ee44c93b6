#!/usr/bin/env python3
"""profiles/valu_<wl>.json from an SQ-counter summary (tools/pmc_summarize.py): VALU instructions per state
and step, and the share of wave-cycles that issue / wait, summed over the kernels of a Jacobian step.
usage: valu_roof.py <sq_counters.json> <states per launch> <launches per kernel name:count,...> <label> [library file name]"""
import json
import sys

d = json.load(open(sys.argv[1]))
n = int(sys.argv[2])
kernels = dict((kv.split(':')[0], int(kv.split(':')[1])) for kv in sys.argv[3].split(','))
tot = dict(valu=0.0, any=0.0, cyc=0.0, wait=0.0, salu=0.0)
per = {}
for name, cnt in kernels.items():
    k = next(v for key, v in d.items() if key.startswith(name))
    m = lambda c: k[c]['mean'] if c in k else 0.0
    # counters are sums over the wavefronts of one launch; SQ_INSTS_VALU counts wavefront instructions
    per[name] = dict(launches_per_step=cnt, valu_wave_instr=m('SQ_INSTS_VALU'), salu_wave_instr=m('SQ_INSTS_SALU'),
                     issue_frac=m('SQ_ACTIVE_INST_ANY') / m('SQ_WAVE_CYCLES'), wait_frac=m('SQ_WAIT_INST_ANY') / m('SQ_WAVE_CYCLES'))
    tot['valu'] += cnt * m('SQ_INSTS_VALU'); tot['salu'] += cnt * m('SQ_INSTS_SALU')
    tot['any'] += cnt * m('SQ_ACTIVE_INST_ANY'); tot['cyc'] += cnt * m('SQ_WAVE_CYCLES'); tot['wait'] += cnt * m('SQ_WAIT_INST_ANY')
print(json.dumps(dict(source=sys.argv[4], library=(sys.argv[5] if len(sys.argv) > 5 and sys.argv[5] not in ('', '-') else None), states_per_launch=n,
                      valu_instr_per_state=tot['valu'] / (n / 64.0),      # per lane = per state
                      salu_instr_per_state=tot['salu'] / (n / 64.0), issue_frac=tot['any'] / tot['cyc'],
                      wait_frac=tot['wait'] / tot['cyc'], kernels=per), indent=1))
