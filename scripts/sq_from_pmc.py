"""profiles/<tag>_sq_counters.json from the SQ counter passes of scripts/profile_round.sh: per wf:: kernel, the average
counter value per launch.
    python scripts/sq_from_pmc.py <sq1.csv> <sq2.csv> <out.json>"""
import collections
import csv
import json
import sys

out = collections.OrderedDict()
for path in sys.argv[1:3]:
    tot, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if "wf::" not in name:
            continue
        key = (name, r["Counter_Name"])
        tot[key] += float(r["Counter_Value"])
        n[key] += 1
    for (name, counter), v in tot.items():
        out.setdefault(name, collections.OrderedDict())[counter] = v / n[(name, counter)]
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(f"{len(out)} kernels -> {sys.argv[3]}")
