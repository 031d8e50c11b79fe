"""Identity of the kernel sources a measurement was taken on: sha256 over csrc/*.hip, common.h and include/x3dhip.h (first
16 hex digits) -- computable on the GPU box, where there is no .git -- plus the git commit when available.  The PMC
collections under profiles/ carry it; bench.py drops a collection whose stamp differs from the tree it runs on."""
import glob
import hashlib
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha16():
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "x3d-multigrid_amd", "csrc", "*.hip"))) + \
        [os.path.join(ROOT, "x3d-multigrid_amd", "csrc", "common.h"), os.path.join(ROOT, "include", "x3dhip.h")]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def commit():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                              stdin=subprocess.DEVNULL, timeout=10).stdout.strip() or None
    except Exception:
        return None


def meta():
    return {"csrc_sha16": csrc_sha16(), "commit": commit()}
