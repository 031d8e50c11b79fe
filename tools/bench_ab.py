"""Print value / ms_per_step of bench.py JSON lines saved to files:  python tools/bench_ab.py a.json b.json ..."""
import json
import sys

for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print("%-40s %8.2f %s  %7.3f ms/step" % (f, d["value"], d["unit"], d["ms_per_step"]))
