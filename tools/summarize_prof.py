"""Condense rocprofv3 CSV output (gpurun_out/prof/...) into small committed summaries under
profiles/: per-kernel duration stats and HBM traffic per launch from the PMC passes.

gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports exactly half the bytes of a wide
coalesced streaming read -> doubled here; WRITE_SIZE is exact for 16-B-per-lane streaming stores.
Both counters are in KiB.
"""
import csv
import glob
import json
import os
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
tag = sys.argv[2] if len(sys.argv) > 2 else "r02"
os.makedirs("profiles", exist_ok=True)


def newest(pattern):
    """gpurun merges each call's files into gpurun_out/ without removing older ones: take the latest."""
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1:]


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    return name if len(name) < 100 else name[:97] + "..."


stats = newest(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))
rows = []
if stats:
    for r in csv.DictReader(open(stats[0])):
        rows.append({"kernel": short(r["Name"]), "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                     "min_ns": int(r["MinNs"]), "max_ns": int(r["MaxNs"]), "pct": float(r["Percentage"])})
with open("profiles/%s_kernel_stats.md" % tag, "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --no-extras\n\n")
    f.write("| kernel | calls | avg us | min us | max us | % |\n|---|---|---|---|---|---|\n")
    for r in rows:
        f.write("| `%s` | %d | %.1f | %.1f | %.1f | %.2f |\n" % (r["kernel"], r["calls"], r["avg_ns"] / 1e3,
                                                                r["min_ns"] / 1e3, r["max_ns"] / 1e3, r["pct"]))

gstats = newest(os.path.join(src, "gemm_stats", "*", "*_kernel_stats.csv"))
if gstats:
    with open("profiles/%s_gemm_kernel_stats.md" % tag, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python3 tools/gemm_bench.py 8192x8192x8192 nn\n\n")
        f.write("2 * 8192^3 = 1.0995e12 FLOP per launch\n\n")
        f.write("| kernel | calls | avg ms | min ms | TFLOP/s at avg |\n|---|---|---|---|---|\n")
        for r in csv.DictReader(open(gstats[0])):
            if "gemm_" not in r["Name"]:
                continue
            avg = float(r["AverageNs"])
            f.write("| `%s` | %d | %.3f | %.3f | %.1f |\n" % (short(r["Name"]), int(r["Calls"]), avg / 1e6,
                                                              int(r["MinNs"]) / 1e6, 2 * 8192.0 ** 3 / avg / 1e3))
    print(open("profiles/%s_gemm_kernel_stats.md" % tag).read())

pmc = {}
for which, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    files = newest(os.path.join(src, which, "*", "*_counter_collection.csv"))
    if not files:
        continue
    vals = {}
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] == ctr and "cast_kernel" in r["Kernel_Name"]:
            vals.setdefault(short(r["Kernel_Name"]), []).append(float(r["Counter_Value"]))
    for k, v in vals.items():
        pmc.setdefault(k, {})[ctr + "_KiB_avg"] = sum(v) / len(v)
        pmc[k]["launches_" + ctr] = len(v)
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 3 --warmup 1 --no-extras",
       "correction": "FETCH_SIZE x2 on gfx950 (wide coalesced reads are tallied at half), WRITE_SIZE exact; KiB -> bytes x1024",
       "kernels": pmc}
for k, d in pmc.items():
    if "FETCH_SIZE_KiB_avg" in d and "WRITE_SIZE_KiB_avg" in d:
        d["hbm_read_bytes"] = 2 * d["FETCH_SIZE_KiB_avg"] * 1024
        d["hbm_write_bytes"] = d["WRITE_SIZE_KiB_avg"] * 1024
        out["hbm_bytes_per_launch"] = d["hbm_read_bytes"] + d["hbm_write_bytes"]
        out["dominant_kernel"] = k
import hashlib

_so = os.path.join("fp8-emulation-toolkit_amd", "mpemu", "pytquant", "hip", "libfpemu_hip.so")
if os.path.isfile(_so):  # bench.py reports this traffic only while the library it loads is this build
    out["lib_sha256"] = hashlib.sha256(open(_so, "rb").read()).hexdigest()
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
print(open("profiles/%s_kernel_stats.md" % tag).read()[:1500])
