#!/usr/bin/env python3
"""Per-kernel duration (min / median, ns) and the gap before each launch from a rocprofv3 --kernel-trace results .db
   python tools/kernel_durations.py gpurun_out/prof_q/q_results.db"""
import collections, sqlite3, sys
import numpy as np
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kt = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = c.execute(f"select s.kernel_name, d.start, d.end from {kt} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
D = collections.defaultdict(list); prev = None
for n, s, e in rows:
    D[n[:48]].append((e - s, (s - prev) if prev else 0)); prev = e
for n, v in D.items():
    d = np.array([x[0] for x in v]); g = np.array([x[1] for x in v])
    print(f"{n:48s} calls {len(v):4d}  duration min {d.min():9d} median {int(np.median(d)):9d}  gap before (median) {int(np.median(g)):7d}")
