#!/usr/bin/env python3
"""profiles/summarize.py <tag> -- condense gpurun_out/prof_<tag>/ (rocprofv3 CSVs) into profiles/<tag>/.

Writes kernel_stats.csv (the --kernel-trace --stats table, verbatim) and pmc_summary.json (per kernel: mean
counter values per launch, plus derived HBM traffic per launch with the gfx950 FETCH_SIZE correction of
MI355X_MICROARCH.md: FETCH_SIZE counts 64 B per 128-B request, i.e. half of the bytes of coalesced reads)."""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles", tag)
os.makedirs(dst, exist_ok=True)
ks = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
if ks:
    shutil.copy(ks[0], os.path.join(dst, "kernel_stats.csv"))
out = collections.defaultdict(dict)
for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    fs = glob.glob(os.path.join(src, d, "*", "*_counter_collection.csv"))
    if not fs:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "rtmi::" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
            out[r["Kernel_Name"]]["launch_shape"] = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "VGPR_Count", "SGPR_Count") if k in r}
    for (k, c), v in agg.items():
        out[k][c] = {"launches": len(v), "mean_per_launch": sum(v) / len(v)}
for k, c in out.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        fetch = c["FETCH_SIZE"]["mean_per_launch"] * 1024 * 2  # gfx950: doubled (see module docstring)
        write = c["WRITE_SIZE"]["mean_per_launch"] * 1024
        c["hbm_bytes_per_launch"] = {"read_corrected": fetch, "write": write, "total": fetch + write}
# which build these counters belong to (bench.py quotes it next to figures it did not measure itself)
try:
    import subprocess
    head = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], text=True).strip()
    dirty = subprocess.call(["git", "-C", root, "diff", "--quiet", "--", "raytracerfortran_amd/csrc"]) != 0
    out["build_commit"] = head + (" + uncommitted changes to csrc/" if dirty else "")
except Exception:
    out["build_commit"] = "unknown"
json.dump(out, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
