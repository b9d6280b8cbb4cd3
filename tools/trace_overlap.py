#!/usr/bin/env python3
"""Concurrency in a rocprofv3 --kernel-trace CSV (last N launches): summed kernel time, time with at least one kernel
running (union), wall span, time during which two or more kernels ran, idle share, and the largest (running kernel,
kernel started beside it) pairs.  usage: python tools/trace_overlap.py <kernel_trace.csv> [N=4000]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-(int(sys.argv[2]) if len(sys.argv) > 2 else 4000):]
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), 1))
    ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
depth, last, union, multi = 0, ev[0][0], 0, 0
for t, d in ev:
    if depth >= 1:
        union += t - last
    if depth >= 2:
        multi += t - last
    depth += d
    last = t
span = ev[-1][0] - ev[0][0]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
print(f"launches {len(rows)}  wall span {span / 1e6:.2f} ms  summed kernel time {busy / 1e6:.2f} ms  >=1 kernel running "
      f"{union / 1e6:.2f} ms  >=2 running {multi / 1e6:.2f} ms  idle {100 * (1 - union / span):.1f} %  "
      f"queues {sorted(set(r.get('Queue_Id') for r in rows))}")
pairs, prev_end, prev = {}, 0, None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < prev_end:
        k = (prev["Kernel_Name"][:60], r["Kernel_Name"][:60])
        pairs[k] = pairs.get(k, 0) + min(e, prev_end) - s
    if e > prev_end:
        prev_end, prev = e, r
for k, v in sorted(pairs.items(), key=lambda kv: -kv[1])[:8]:
    print(f"{v / 1e3:9.1f} us  {k[0]}  ||  {k[1]}")
