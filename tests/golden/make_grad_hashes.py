"""Regenerate tests/golden/grad_hashes.json (run on an MI355X box whenever csrc/ changed):

    python tests/golden/make_grad_hashes.py [out.json]        # default: gpurun_out/grad_hashes.json, copy it to tests/golden/

sha256 over logits + all parameter gradients of four small golden training cases on the current kernel sources, next to the
plain-bound margins of that very result (so tests/parity.py may hold a pinned result to the PLAIN bounds) and the identity
of the box that computed it."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "x3d-multigrid_amd"))

import torch  # noqa: E402

from tests import gradhash, parity  # noqa: E402
from x3dhip import synthetic  # noqa: E402


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "grad_hashes.json")
    dev = torch.device("cuda:0")
    rec = {"csrc_sha16": gradhash.csrc_sha16(), "box": gradhash.box_identity(), "cases": {}}
    for case in gradhash.CASES:
        first = None
        for rep in range(2):                      # twice in one process: the record itself must be reproducible
            h, logits, loss, grads, g = gradhash.run_case(case, dev)
            assert first is None or h == first, "case %s is not bitwise reproducible in one process" % case
            first = h
        r = parity.grad_report(grads, g, synthetic.gradient_sketch)
        rec["cases"][case] = {"sha256": h, "global_norm_err": r["global_norm_err"], "norm_err_max": r["norm_err_max"],
                              "sketch_err": r["sketch_err"], "logits_err": parity.rel(logits, g["logits"])}
        print(case, h[:16], "global norm err %.2e" % r["global_norm_err"], flush=True)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(rec, open(out, "w"), indent=1, sort_keys=True)
    print("wrote", out)


if __name__ == "__main__":
    main()
