#!/usr/bin/env python3
"""Gaps between consecutive kernels of a rocprofv3 kernel trace (csv): python scripts/kernel_gaps.py <kernel_trace.csv>
Prints the total kernel time, the total idle time between kernels inside the busiest window, and the largest gaps."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in rows), key=lambda e: e[0])
# steady state: from the third-last GAE launch (one per PPO iteration) to the last Adam launch
gae = [i for i, e in enumerate(ev) if "ds_gae_partial_kernel" in e[2]]
adam = [i for i, e in enumerate(ev) if "adam_kernel" in e[2]]
if len(gae) >= 3 and adam:
    ev = ev[gae[-3]:adam[-1] + 1]
else:
    ev = ev[len(ev) * 2 // 3:]
busy = sum(e - s for s, e, _ in ev)
span = ev[-1][1] - ev[0][0]
gaps = [(ev[i + 1][0] - ev[i][1], ev[i][2], ev[i + 1][2]) for i in range(len(ev) - 1)]
pos = [g for g in gaps if g[0] > 0]
print("kernels %d  span %.3f ms  busy %.3f ms  idle %.3f ms (%.1f %%)  overlapped pairs %d" %
      (len(ev), span / 1e6, busy / 1e6, (span - busy) / 1e6, 100.0 * (span - busy) / span, sum(1 for g in gaps if g[0] < 0)))
import collections
hist = collections.Counter(min(int(g[0] / 1000), 20) for g in pos)
print("gap histogram (us: count):", sorted(hist.items()))
by = collections.defaultdict(list)
for g, a, b in pos:
    by[(a[:40], b[:40])].append(g)
print("largest mean gaps by kernel pair:")
for (a, b), v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print("  %6.2f us x %4d = %7.1f us   %s -> %s" % (sum(v) / len(v) / 1e3, len(v), sum(v) / 1e3, a, b))

# the glue between the rollout graph and the update graphs: from the GAE launch to the first Adam launch after it
names = [e[2] for e in ev]
for gi in [i for i, nme in enumerate(names) if "ds_gae_partial_kernel" in nme][:3]:
    aj = next((j for j in range(gi, len(ev)) if "adam_kernel" in names[j]), None)
    if aj is None:
        break
    seg = ev[gi:aj]
    print("glue: %d kernels, span %.1f us, busy %.1f us" % (len(seg), (ev[aj][0] - ev[gi][0]) / 1e3, sum(e - s for s, e, _ in seg) / 1e3))
    if "-v" in sys.argv:
        for s_, e_, n_ in seg:
            print("    %6.1f us  %s" % ((e_ - s_) / 1e3, n_))
        break
