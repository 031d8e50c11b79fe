"""Turn rocprofv3 PMC passes (FETCH_SIZE pass + WRITE_SIZE pass, each with --kernel-trace only) of
`python3 bench.py --no-graph ...` into per-launch HBM traffic per kernel family, applying the gfx950
correction from /opt/skills/guides/MI355X_MICROARCH.md (section HBM): FETCH_SIZE under-reports wide coalesced
reads by exactly 2x; both counters are in KiB.

    python tools/collect_traffic.py <fetch_dir> <write_dir> <out.json>
"""
import collections
import csv
import glob
import json
import re
import sys

FAMILY = [("pw_wgrad", "pw_bwd_weight"),
          ("dw_fwd_kernel", "dw333_fwd"), ("dw_bwd_kernel", "dw333_bwd"), ("bn_stats_add_relu_fwd", "bn_stats_add_relu_fwd"), ("bn_add_relu_fwd", "bn_add_relu_fwd"),
          ("bn_add_relu_bwd", "bn_add_relu_bwd"), ("dw5t_fwd", "dw5t_fwd"), ("dw5t_bwd", "dw5t_bwd"),
          ("stem133_fwd", "stem133_fwd"), ("stem133_wgrad", "stem133_bwd_weight")]


def load(d, counter):
    """rocprofv3 7.2 writes a rocpd sqlite database by default (view counters_collection); older
    --output-format csv runs leave *counter_collection.csv."""
    per = collections.defaultdict(list)
    dbs = glob.glob(d + "/*.db") + glob.glob(d + "/*/*.db")
    if dbs:
        import sqlite3
        con = sqlite3.connect(dbs[0])
        for name, val in con.execute("select kernel_name, value from counters_collection where counter_name=?", (counter,)):
            per[name].append(float(val))
        return per
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            per[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return per


def fam(name):
    """bench.py op family of a kernel name.  Pointwise forward and data-gradient share kernel templates: the input-mode
    template argument tells them apart (IN = 2 is the BN-backward prologue of the data gradient)."""
    if "pw_bwd_fused_kernel" in name:
        return "pw_bwd_fused"
    if "pw6_kernel" in name or "pw8_kernel" in name or "pw_fwd_stream_kernel" in name:
        return "pw_fwd"
    if "pw7_kernel" in name or "pw7r_kernel" in name or "pw5_kernel" in name:
        return "pw_bwd_data"
    m = re.search(r"pw4_kernel<(\d+)", name) or re.search(r"pw2_kernel<(\d+)", name)
    if m:
        return "pw_bwd_data" if m.group(1) == "2" else "pw_fwd"
    m = re.search(r"pw3_kernel<\d+, \d+, (\d+)", name) or re.search(r"::pw_kernel<\d+, \d+, (\d+)", name)
    if m:
        return "pw_bwd_data" if m.group(1) == "2" else "pw_fwd"
    for key, f in FAMILY:
        if key in name:
            return f
    return None


def template_name(name):
    """Kernel template without its arguments and namespace: what x3d_last_kernel() / bench.py's roofline.kernel report."""
    m = re.search(r"([A-Za-z_][A-Za-z_0-9]*)(?:<[^(]*>)?\(", name)
    return m.group(1) if m else name.split("(")[0]


def main():
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import stamp
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
    kagg = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for k, vals in fetch.items():
        for key, table in ((fam(k), agg), (template_name(k), kagg)):
            if key:
                table[key][0] += 2.0 * 1024.0 * sum(vals)          # x2: gfx950 FETCH_SIZE correction, KiB -> B
                table[key][2] += len(vals)
    for k, vals in write.items():
        for key, table in ((fam(k), agg), (template_name(k), kagg)):
            if key:
                table[key][1] += 1024.0 * sum(vals)
    row = lambda rd, wr, n: {"launches": n, "hbm_read_bytes_per_launch": rd / max(n, 1),
                             "hbm_write_bytes_per_launch": wr / max(n, 1), "hbm_bytes_per_launch": (rd + wr) / max(n, 1)}
    out = {f: row(*v) for f, v in agg.items()}
    out["_kernels"] = {k: row(*v) for k, v in kagg.items() if "Cijk" not in k and "rocclr" not in k}
    out["_meta"] = stamp.meta()
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    del out["_kernels"], out["_meta"]
    for f, v in sorted(out.items()):
        print("%-20s launches %5d  read %10.1f MB  write %10.1f MB per launch" %
              (f, v["launches"], v["hbm_read_bytes_per_launch"] / 1e6, v["hbm_write_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
