"""Per-kernel-family sums of SQ counters from one rocprofv3 --pmc pass of `python3 bench.py --no-graph ...`
(the pass carries --kernel-trace only, as the pool requires), plus the ratios the round-2 review asks for:

  depthwise stencils : SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (share of LDS cycles lost to bank conflicts),
                       SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES, SQ_INSTS_VALU per launch
  pointwise GEMMs    : SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES summed over the SEs) = MFMA utilisation,
                       SQ_INSTS_MFMA per launch

    python tools/collect_pmc.py <pass_dir> [<pass_dir> ...] <out.json>
"""
import collections
import glob
import json
import sqlite3
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from collect_traffic import fam, template_name  # noqa: E402
import stamp  # noqa: E402


def load_all(d):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for db in glob.glob(d + "/*.db") + glob.glob(d + "/*/*.db"):
        con = sqlite3.connect(db)
        for name, cname, val in con.execute("select kernel_name, counter_name, value from counters_collection"):
            per[cname][name].append(float(val))
    return per


def main():
    dirs, out_path = sys.argv[1:-1], sys.argv[-1]
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for d in dirs:
        for cname, kernels in load_all(d).items():
            for k, vals in kernels.items():
                f = fam(k) or ("finalize/other" if "bn_" in k or "se_" in k or "reduce" in k else None)
                if f:
                    agg[f][cname][0] += sum(vals)
                    agg[f][cname][1] += len(vals)
                if "Cijk" not in k and "rocclr" not in k:
                    agg["kernel:" + template_name(k)][cname][0] += sum(vals)
                    agg["kernel:" + template_name(k)][cname][1] += len(vals)
    out = {}
    for f, cs in sorted(agg.items()):
        row = {c: {"sum": v[0], "launches": v[1], "per_launch": v[0] / max(v[1], 1)} for c, v in cs.items()}
        g = lambda c: cs[c][0] if c in cs else None
        der = {}
        if g("SQ_LDS_BANK_CONFLICT") is not None and g("SQ_LDS_IDX_ACTIVE"):
            der["lds_bank_conflict_per_idx_active"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
        if g("SQ_WAIT_INST_LDS") is not None and g("SQ_WAVE_CYCLES"):
            der["wait_inst_lds_per_wave_cycle"] = g("SQ_WAIT_INST_LDS") / g("SQ_WAVE_CYCLES")
        if g("SQ_VALU_MFMA_BUSY_CYCLES") is not None and g("SQ_BUSY_CYCLES"):
            # SQ_VALU_MFMA_BUSY_CYCLES counts cycles, SQ_BUSY_CYCLES quad-cycles per SE (MI355X_MICROARCH.md, cycle constants)
            der["mfma_busy_per_sq_busy"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / (4.0 * g("SQ_BUSY_CYCLES"))
        if g("SQ_ACTIVE_INST_VALU") is not None and g("SQ_WAVE_CYCLES"):
            der["active_inst_valu_per_wave_cycle"] = g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES")
        out[f] = {"counters": row, "derived": der}
    out["_meta"] = stamp.meta()
    json.dump(out, open(out_path, "w"), indent=1)
    del out["_meta"]
    for f, v in out.items():
        print("%-18s %s" % (f, "  ".join("%s=%.4g" % kv for kv in v["derived"].items())))
        print("                   " + "  ".join("%s/launch=%.4g" % (c, r["per_launch"]) for c, r in sorted(v["counters"].items())))


if __name__ == "__main__":
    main()
