#!/usr/bin/env python3
"""rocprofv3 --pmc CSV output of two passes (FETCH_SIZE, WRITE_SIZE) -> JSON of HBM-side bytes per launch per kernel.
    python tools/pmc_to_json.py <workload> <scenes_per_launch> <fetch_dir> <write_dir>
Corrections as MI355X_MICROARCH.md (HBM section) prescribes for gfx950: FETCH_SIZE tallies 128-B requests of wide
coalesced streaming reads at 64 B -> the read side is doubled; WRITE_SIZE is exact for 16-B-per-lane stores. Units: KB."""
import collections, csv, glob, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gencomm_amd import _lib

workload, B, fdir, wdir = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]


def load(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


fetch, write = load(fdir, "FETCH_SIZE"), load(wdir, "WRITE_SIZE")
out = {"_about": "HBM-side traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --steps 2 --warmup 1 "
                 "--streams 1 --no-cpu-baseline --no-exact --no-timer` (tools/pmc_pass.sh). hbm_bytes_per_launch = (2 x FETCH_SIZE + "
                 "WRITE_SIZE) x 1024: on gfx950 FETCH_SIZE counts the 128-B requests of wide coalesced reads as 64 B "
                 "(MI355X_MICROARCH.md, HBM section), WRITE_SIZE is exact for 16-B-per-lane stores; Infinity-Cache hits are included "
                 "(the counters sit on the L2's fabric side). raw = FETCH_SIZE + WRITE_SIZE without the doubling.",
       "workload": workload, "scenes_per_launch": B, "library_src": _lib.library_src_hash(), "kernels": {}}
fam_f, fam_w = [], []
for name in sorted(set(fetch) | set(write)):
    f, w = fetch.get(name, []), write.get(name, [])
    if not f or not w:
        continue
    fm, wm = sum(f) / len(f), sum(w) / len(w)
    out["kernels"][name] = {"launches_sampled": min(len(f), len(w)), "FETCH_SIZE_KB_mean": round(fm, 1), "WRITE_SIZE_KB_mean": round(wm, 1),
                            "hbm_bytes_per_launch_raw": int((fm + wm) * 1024), "hbm_bytes_per_launch": int((2 * fm + wm) * 1024)}
    if re.search(r"conv8h_kernel<", name):
        fam_f += f
        fam_w += w
if fam_f and fam_w:
    fm, wm = sum(fam_f) / len(fam_f), sum(fam_w) / len(fam_w)
    out["conv8h_family"] = {"launches_sampled": min(len(fam_f), len(fam_w)), "FETCH_SIZE_KB_mean": round(fm, 1), "WRITE_SIZE_KB_mean": round(wm, 1),
                            "hbm_bytes_per_launch_raw": int((fm + wm) * 1024), "hbm_bytes_per_launch": int((2 * fm + wm) * 1024),
                            "note": "mean over every conv8h_kernel<...> launch of the run (all variants, full- and half-resolution levels): the "
                                    "same launch mix bench.py's roofline averages over"}
print(json.dumps(out, indent=1))
