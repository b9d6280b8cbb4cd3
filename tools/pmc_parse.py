"""FETCH_SIZE / WRITE_SIZE passes of tools/pmc_kernels.py -> profiles/r02_pmc_traffic.{json,txt}.
    python tools/pmc_parse.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass>
Counter unit: KB.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies the 128-byte requests of
16-byte-per-lane streaming reads at 64 bytes -> doubled; WRITE_SIZE is exact for 16-byte stores."""
import collections
import csv
import glob
import json
import re
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_dispatch(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    out = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        key = (int(r["Dispatch_Id"]), r["Kernel_Name"], r.get("Grid_Size", ""))
        out[key] = out.get(key, 0.0) + float(r["Counter_Value"])
    return out


def main():
    rd, wr = per_dispatch(sys.argv[1], "FETCH_SIZE"), per_dispatch(sys.argv[2], "WRITE_SIZE")

    def groups(dd):
        g = collections.OrderedDict()
        for (did, name, grid), v in dd.items():
            if "lstm_rs_fwd" in name or "lstm_cluster_fwd" in name or "conv3x3_s1_kernel" in name or "conv3x3_c32_kernel" in name:
                short = re.sub(r"\(anonymous namespace\)::|^void ", "", name)
                short = short[:short.rfind("(")] if "(" in short else short
                g.setdefault((short, grid), []).append(v)
        return g
    gr, gw = groups(rd), groups(wr)
    # launches appear in program order: recurrent kernel, then convs at B = 16 (64->64, 32->32), then B = 64
    labels = {}
    convs = [k for k in gr if "conv3x3" in k[0]]
    names = ["conv3x3_s1_B16", "conv3x3_c32_B16", "conv3x3_s1_B64", "conv3x3_c32_B64"]
    for k, n in zip(convs, names):
        labels[k] = n
    for k in gr:
        if "lstm_rs_fwd" in k[0] or "lstm_cluster_fwd" in k[0]:
            labels[k] = "lstm_fwd_H512_N2080_L16"
    res, lines = {}, []
    for k, lab in labels.items():
        r = sum(gr[k]) / len(gr[k]) * 1024 * 2
        w = sum(gw[k]) / len(gw[k]) * 1024 if k in gw else float("nan")
        res[lab] = {"kernel": k[0], "grid": k[1], "fetch_bytes_x2": r, "write_bytes": w, "hbm_bytes_per_launch": r + w,
                    "launches": len(gr[k])}
        lines.append(f"{lab:28s} {k[0][:60]:60s} grid {k[1]:>9s}  read {r / 1e6:9.1f} MB (FETCH_SIZE x2)  "
                     f"written {w / 1e6:9.1f} MB  total {(r + w) / 1e6:9.1f} MB  ({len(gr[k])} launches)")
    with open(os.path.join(ROOT, "profiles", "r02_pmc_traffic.json"), "w") as f:
        json.dump(res, f, indent=1)
    with open(os.path.join(ROOT, "profiles", "r02_pmc_traffic.txt"), "w") as f:
        f.write(__doc__ + "\n" + "\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
