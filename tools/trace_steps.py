#!/usr/bin/env python3
"""Per-kernel time of the LAST K training steps of a rocprofv3 --kernel-trace CSV: steps are delimited by
adamw_update_kernel, which runs once per step.  Of a bench.py run these are the eagerly issued one-stream steps of its
roofline probe (a HIP-graph replay's kernels carry no per-step order a trace could be cut at).

    python tools/trace_steps.py <kernel_trace.csv> [K] [rows]
"""
import collections
import csv
import re
import sys


def main():
    path = sys.argv[1]
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    rows_out = int(sys.argv[3]) if len(sys.argv) > 3 else 45
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "adamw_update_kernel" in r["Kernel_Name"]]
    # whole steps only: an optimizer launch that closes at least 100 launches of forward + backward (bench.py's
    # closing clip+AdamW timing calls follow each other directly)
    marks = [m for j, m in enumerate(marks) if j > 0 and m - marks[j - 1] > 100]
    if len(marks) < K + 1:
        raise SystemExit(f"only {len(marks)} optimizer steps in the trace")
    lo, hi = marks[-K - 1] + 1, marks[-1] + 1
    sel = rows[lo:hi]
    span = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e6 / K
    agg = collections.defaultdict(lambda: [0, 0])
    for r in sel:
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        n = re.sub(r"^void ", "", n)
        g = f' grid {r["Grid_Size_X"]}x{r["Grid_Size_Y"]}x{r["Grid_Size_Z"]}' if "--grid" in sys.argv else ""
        agg[n[:110] + g][0] += 1
        agg[n[:110] + g][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot = sum(v[1] for v in agg.values()) / 1e6 / K
    print(f"steps {K}  launches/step {len(sel) / K:.0f}  span/step {span:.3f} ms  kernel-busy/step {tot:.3f} ms  "
          f"idle {100 * (1 - tot / span):.1f} %")
    print(f"{'kernel':112s} {'n/step':>7s} {'ms/step':>9s} {'avg us':>9s} {'%':>6s}")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:rows_out]:
        print(f"{n:112s} {c / K:7.1f} {t / 1e6 / K:9.3f} {t / c / 1e3:9.1f} {100 * t / 1e6 / K / tot:6.2f}")


if __name__ == "__main__":
    main()
