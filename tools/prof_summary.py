"""Shared summarisers of rocprofv3 CSV output for tools/gpu.sh.

  prof_summary.py kstats DIR [KEYWORD ...] [--per N] [--top N]   per-kernel table of DIR/**/_kernel_stats.csv; with
                                                                  KEYWORDs also the total of the kernels matching any
  prof_summary.py pmc DIR KERNEL_SUBSTR                           average of every counter in DIR/**/_counter_collection.csv
                                                                  over the launches of the kernels matching KERNEL_SUBSTR
  prof_summary.py hbm DIR                                         FETCH_SIZE / WRITE_SIZE per kernel (KiB per launch)
  prof_summary.py gaps DIR KEYWORD [KEYWORD ...]                  per position of a repeating launch sequence: duration
                                                                  and the gap to the next kernel (kernel_trace.csv)
"""
import collections
import csv
import glob
import sys


def newest(root, suffix):
    files = sorted(glob.glob(root + "/**/*" + suffix, recursive=True))
    if not files:
        sys.exit("no *%s under %s" % (suffix, root))
    return files[-1]


def kstats(root, keys, per, top):
    rows = list(csv.DictReader(open(newest(root, "_kernel_stats.csv"))))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("all kernels: %.3f ms%s" % (tot / 1e6 / per, "" if per == 1 else " per unit (%d units)" % per))
    if keys:
        ours = [r for r in rows if any(k in r["Name"] for k in keys)]
        print("kernels matching %s: %.3f ms, %d launches%s" % (
            "/".join(keys), sum(float(r["TotalDurationNs"]) for r in ours) / 1e6 / per,
            sum(int(r["Calls"]) for r in ours) // per, "" if per == 1 else " per unit"))
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    for r in rows[:top]:
        print("| %s | %s | %.2f | %.1f | %.1f | %.1f | %s |" % (
            r["Name"].replace("(anonymous namespace)::", "")[:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
            float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))


def pmc(root, kernel):
    acc = collections.defaultdict(list)
    for f in glob.glob(root + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    v = {k: sum(x) / len(x) for k, x in acc.items()}
    for k in sorted(v):
        print("%-26s %.4g per launch (%d launches)" % (k, v[k], len(acc[k])))
    w = v.get("SQ_WAVE_CYCLES")
    if w:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
                  "SQ_ACTIVE_INST_LDS"):
            if k in v:
                print("%-26s %.1f %% of wave cycles" % (k, 100 * v[k] / w))
    if v.get("SQ_LDS_IDX_ACTIVE"):
        print("LDS bank conflicts         %.2f %% of LDS-array cycles" % (100 * v.get("SQ_LDS_BANK_CONFLICT", 0) / v["SQ_LDS_IDX_ACTIVE"]))


def hbm(root):
    """KiB per launch; on gfx950 FETCH_SIZE tallies wide coalesced reads at half (x2 for bytes), WRITE_SIZE is exact."""
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        acc = collections.defaultdict(list)
        for f in glob.glob(root + "/**/*_counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == ctr:
                    acc[r["Kernel_Name"].replace("(anonymous namespace)::", "")[:90]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print("%s %s: avg %.1f KiB over %d launches" % (ctr, k, sum(v) / len(v), len(v)))


def gaps(root, keys):
    rows = [r for r in csv.DictReader(open(newest(root, "kernel_trace.csv"))) if any(k in r["Kernel_Name"] for k in keys)]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[len(rows) // 2:]                       # the steady half
    names = [next(k for k in keys if k in r["Kernel_Name"]) for r in rows]
    last = keys[-1]                                    # the sequence ends with the last keyword's kernel
    first = names.index(last) + 1
    rows, names = rows[first:], names[first:]
    per = names.index(last) + 1
    acc = {}
    for i, r in enumerate(rows):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a = acc.setdefault((i % per, names[i]), [[], []])
        a[0].append(d)
        if i + 1 < len(rows):
            a[1].append((int(rows[i + 1]["Start_Timestamp"]) - int(r["End_Timestamp"])) / 1e3)
    tot = 0.0
    for (pos, nm), (d, g) in sorted(acc.items()):
        print("  %d %-16s dur %8.1f us   gap to next %6.1f us" % (pos, nm, sum(d) / len(d), sum(g) / max(1, len(g))))
        tot += sum(d) / len(d) + sum(g) / max(1, len(g))
    print("  per sequence: %.1f us" % tot, flush=True)


if __name__ == "__main__":
    cmd, root, rest = sys.argv[1], sys.argv[2], sys.argv[3:]
    if cmd == "kstats":
        per = top = None
        keys = []
        it = iter(rest)
        for a in it:
            if a == "--per":
                per = int(next(it))
            elif a == "--top":
                top = int(next(it))
            else:
                keys.append(a)
        kstats(root, keys, per or 1, top or 25)
    elif cmd == "pmc":
        pmc(root, rest[0])
    elif cmd == "hbm":
        hbm(root)
    elif cmd == "gaps":
        gaps(root, rest)
    else:
        sys.exit(__doc__)
