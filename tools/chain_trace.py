"""Ordered kernel timeline of ONE replayed training step from a rocprofv3 --kernel-trace run of bench.py (rocpd database):
start offset, duration and gap to the previous kernel of every launch, cut into stem / layerN.blockM / head segments by the
schedule's own marker kernels, plus per-segment and per-class sums.

    rocprofv3 --kernel-trace -d <dir> -- python3 bench.py --no-cpu-baseline --no-exact-fp32 --no-kernel-timing --steps 6
    python tools/chain_trace.py <dir> [out.txt]

The step is found as the LAST run of launches between two `pw_pack_batch_kernel` launches (the first launch of every
forward pass: engine.WeightPacks.refresh).
"""
import glob
import re
import sqlite3
import sys


def short(name):
    m = re.search(r"::(\w+)(<[^>]*>)?", name)
    if m:
        return m.group(1) + (m.group(2) or "")
    return name.split("(")[0][:60]


FINALIZE = ("bn_fwd_fused", "bn_bwd_fused", "se_fwd", "se_bn_fwd", "se_bwd", "se_tail", "reduce_tiles", "reduce_partials", "bn_eval")


def klass(n):
    if any(k in n for k in FINALIZE):
        return "finalize"
    if "dw_fwd" in n or "dw_bwd" in n:
        return "depthwise"
    if "wgrad" in n:
        return "wgrad"
    if n.startswith("pw"):
        return "pointwise"
    if "bn_" in n:
        return "elementwise"
    return "other"


def load(d):
    dbs = glob.glob(d + "/*.db") + glob.glob(d + "/*/*.db") + glob.glob(d + "/*/*/*.db")
    con = sqlite3.connect(dbs[0])
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    name_c = "name" if "name" in cols else "kernel_name"
    rows = list(con.execute("select %s, start, end from kernels order by start" % name_c))
    return [(short(n), int(s), int(e)) for n, s, e in rows]


def main():
    rows = load(sys.argv[1])
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    packs = [i for i, r in enumerate(rows) if r[0].startswith("pw_pack_batch_kernel")]
    if len(packs) < 2:
        raise SystemExit("fewer than two steps in the trace")
    a, b = packs[-2], packs[-1]
    step = rows[a:b]
    t0 = step[0][1]
    span = (step[-1][2] - t0) / 1e3
    busy = sum(e - s for _, s, e in step) / 1e3
    print("# one replayed step: %d launches, span %.1f us, kernel time %.1f us" % (len(step), span, busy), file=out)
    per = {}
    prev_end = t0
    for i, (n, s, e) in enumerate(step):
        k = klass(n)
        c = per.setdefault(k, [0, 0.0])
        c[0] += 1
        c[1] += (e - s) / 1e3
        print("%4d %9.1f %7.2f gap %6.2f  %-10s %s" % (i, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, k, n), file=out)
        prev_end = e
    # the 62 `__amd_rocclr_copyBuffer` per step execution of profiles/r03/h_kernel_stats.csv.  They are
    # not nodes of the replayed graph: count them inside every window between two weight-pack launches (= one execution of
    # the step: eager warm-up, capture, replay) and report which kind of window holds them.
    copies = [i for i, r in enumerate(rows) if "copyBuffer" in r[0] or "copy_buffer" in r[0].lower()]
    per_window = []
    for w0, w1 in zip(packs[:-1], packs[1:]):
        n_all = w1 - w0
        n_cp = len([i for i in copies if w0 <= i < w1])
        per_window.append((n_all, n_cp))
    print("# __amd_rocclr_copyBuffer: %d in the whole trace (%d launches); per step window (launches, copies): %s"
          % (len(copies), len(rows), " ".join("%d/%d" % w for w in per_window)), file=out)
    print("# (windows with %d launches are graph replays; the larger ones are eager warm-up / capture executions)" % len(step), file=out)
    print("# per class:", file=out)
    for k, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print("#   %-12s %4d launches %9.1f us" % (k, c, t), file=out)


if __name__ == "__main__":
    main()
