#!/usr/bin/env python3
"""Median duration of every kernel of the serial leg of a traced bench run, and the median gap in front of it:   python tools/dbg/trace_steps.py <dir with *_kernel_trace.csv>"""
import csv, glob, statistics, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
seq = [(r['Kernel_Name'].split('(')[0][-40:], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows if 'order_' in r['Kernel_Name'] or 'nmpc_ipm_lds' in r['Kernel_Name']]
seq = seq[len(seq) // 8: len(seq) // 2]  # the serial, timed part
d = collections.defaultdict(list); g = collections.defaultdict(list)
pe = None
for n, s, e in seq:
    d[n].append((e - s) / 1e3)
    if pe is not None: g[n].append((s - pe) / 1e3)
    pe = e
for n in d: print(f"{n:42s} n {len(d[n]):4d}  duration median {statistics.median(d[n]):8.2f} us  min {min(d[n]):8.2f}   gap in front median {statistics.median(g[n]) if g[n] else 0:6.2f} us")
