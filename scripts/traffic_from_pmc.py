"""Builds profiles/traffic.json from two rocprofv3 --pmc runs of bench.py (FETCH_SIZE and WRITE_SIZE in separate
passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes).

    python scripts/traffic_from_pmc.py <fetch counter_collection.csv> <write counter_collection.csv> <round tag> [config]
(config: cfg2 -> profiles/traffic.json, else profiles/traffic_<config>.json -- the files bench.py --config reads)

gfx950 corrections (guide, section HBM): WRITE_SIZE is exact for 16-byte-per-lane streaming stores.  FETCH_SIZE
tallies 128-byte requests at 64 bytes, so it is doubled -- for EVERY kernel (rule 3, round 5; `_rule` in the output names the
rule a file was written under, and bench.py carries it into its record so that figures of different rules are not compared).
Calibrations, each a kernel whose compulsory reads are known: evaluate.last_pass must read the 512 MiB intermediate of cfg 2
exactly once (raw counter: 256 MiB); the second strided pass of cfg 3 must read its 17.18 GB exactly once (raw: 8.59 GB);
the three k_merkle_level2 launches of cfg 2 must read 256 + 64 + 16 = 336 MiB of children (raw: 168.4 MiB -- rule 2, round 4,
had left this one kernel undoubled "as rounds 1-3 had it": its Merkle reads were low by half).
When the leaves are hashed inside the last evaluation pass (one segment, one trace) there is no k_hash_rows launch: its
entry is zero and the evaluate entry carries the 256 MiB of leaf writes."""
import collections
import csv
import hashlib
import json
import os
import sys


def csrc_sha():
    """As bench.py: the digest of the kernel sources the counters were collected on."""
    h = hashlib.sha256()
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "starkpack-winterfell_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]

LOGICAL = [
    ("interpolate", ["k_cols_to_seg", "k_seg_strided<wf::F64, 0,", "k_seg_last<wf::F64, 0,", "k_seg_strided<wf::F128, 0,",
                     "k_seg_last<wf::F128, 0,", "k_seg_to_cols", "k_seg_strided_wide<wf::F64, 0,"]),
    ("evaluate", ["k_seg_strided<wf::F64, 1,", "k_seg_last<wf::F64, 1,", "k_seg_last_hash<wf::F64,", "k_seg_strided<wf::F128, 1,",
                  "k_seg_last<wf::F128, 1,", "k_seg_last_hash<wf::F128,", "k_seg_strided_wide<wf::F64, 1,",
                  "k_seg_last_hash_tp<wf::F64,", "k_seg_last_hash_tp<wf::F128,"]),
    ("hash_rows", ["k_hash_rows", "k_hash_chunks", "k_hash_merge_chunks"]),  # (k_hash_chunks also matches k_hash_chunks_staged)
    ("merkle", ["k_merkle_level", "k_merkle_subtree"])  # k_merkle_level also matches k_merkle_level2,
]
# Round 4 calibration: the SECOND strided evaluation pass of cfg 3 must read the 17.18 GB intermediate exactly once (nothing can
# absorb 16 GiB); its raw counter is 8.59 GB -- with 64-byte tile rows (round 3's kernel, same raw figure) as with 128-byte rows.
# So the strided passes are doubled like every other kernel; rounds 1-3 took their 64-byte gathers as "counted exactly" and
# UNDERCOUNTED them by half (cfg 3 evaluate: 87.5 GB then = 104.7 GB by this rule; cfg 2: 2.37 -> 2.64 GB).  The first strided
# pass re-reads the polynomials once per coset at the fabric (cfg 2: 8 x 67 MB) -- Infinity-Cache hits are counted, as the guide says.
NO_DOUBLE = ()  # (rule 2 had k_merkle_level2 here; its raw counter is half its compulsory reads like every other kernel's)
RULE = "3: FETCH_SIZE x 2 for every kernel (round 5); WRITE_SIZE as counted"


def per_kernel(path, counter):
    tot, calls = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or "wf::" not in r["Kernel_Name"]:
            continue
        tot[r["Kernel_Name"]] += float(r["Counter_Value"])
        calls[r["Kernel_Name"]] += 1
    return tot, calls


def main():
    fetch, fcalls = per_kernel(sys.argv[1], "FETCH_SIZE")
    write, _ = per_kernel(sys.argv[2], "WRITE_SIZE")
    steps = fcalls[next(k for k in fcalls if "k_cols_to_seg" in k)]  # one launch per commitment
    config = sys.argv[4] if len(sys.argv) > 4 else "cfg2"
    out = {"_note": f"HBM bytes per commitment ({config}) from rocprofv3 PMC; see scripts/traffic_from_pmc.py for the "
                    "gfx950 corrections", "_round": sys.argv[3], "_steps_profiled": steps, "_csrc_sha": csrc_sha(), "_rule": RULE}
    for name, pats in LOGICAL:
        rd = wr = 0.0
        for k in fetch:
            if any(p in k for p in pats):
                mult = 1.0 if any(n in k for n in NO_DOUBLE) else 2.0
                rd += fetch[k] * 1024 * mult / steps
                wr += write.get(k, 0.0) * 1024 / steps
        out[name] = {"read_bytes_per_step": rd, "write_bytes_per_step": wr, "hbm_bytes_per_step": rd + wr}
    out["path_total"] = sum(v["hbm_bytes_per_step"] for k, v in out.items() if not k.startswith("_"))
    # raw counters per kernel and launch (KiB as rocprofv3 reports them), for the calibration notes
    out["_raw_per_launch_kib"] = {k.split("(")[0][:90]: {"fetch": round(fetch[k] / max(1, fcalls[k]), 1), "write": round(write.get(k, 0.0) / max(1, fcalls[k]), 1),
                                                          "launches_per_step": fcalls[k] / steps} for k in fetch}
    json.dump(out, open("profiles/traffic.json" if config == "cfg2" else f"profiles/traffic_{config}.json", "w"), indent=1)
    print(json.dumps(out, indent=1))


main()
