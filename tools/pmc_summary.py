#!/usr/bin/env python3
"""Summarise rocprofv3 runs of bench.py into the small JSON/CSV files kept under profiles/.

  tools/pmc_summary.py traffic <dir-with-*_counter_collection.csv> <out.json>
      FETCH_SIZE (KB, x2 gfx950 correction for wide streaming reads, MI355X_MICROARCH.md) summed over the kernels of
      one frame-step graph replay (the launches between two k_advance), median over the replays of the run.
  tools/pmc_summary.py mfma <dir> <out.json>
      SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs) per kernel name.
  tools/pmc_summary.py stats <dir-with-*_kernel_stats.csv> <out.csv>      (copies the top of the stats table)
"""
import csv
import glob
import json
import os
import statistics
import sys
from collections import defaultdict


def counter_rows(d):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        sys.exit(f"no *counter_collection.csv under {d}")
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                yield row


def per_dispatch(d):
    disp = {}
    for r in counter_rows(d):
        k = int(r["Dispatch_Id"])
        e = disp.setdefault(k, {"name": r["Kernel_Name"], "c": defaultdict(float)})
        e["c"][r["Counter_Name"]] += float(r["Counter_Value"])
    return [disp[k] for k in sorted(disp)]


def traffic(d, out, source):
    frames, cur = [], None
    for e in per_dispatch(d):
        if cur is not None:
            cur.append(e)
        if e["name"].startswith("k_advance") or "k_advance" in e["name"]:
            if cur:
                frames.append(cur)
            cur = []
    counts = [len(f) for f in frames]
    mode = statistics.mode(counts)
    steps = [f for f in frames if len(f) == mode]
    kb = [sum(e["c"].get("FETCH_SIZE", 0.0) for e in f) for f in steps]
    med = statistics.median(kb)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import csrc_digest              # ties the pass to the kernel sources it measured (bench.py refuses a stale file)
    json.dump({"source": source, "csrc_digest": csrc_digest(), "fetch_size_kb_per_frame_median": med,
               "gfx950_correction": "x2 (MI355X_MICROARCH.md: FETCH_SIZE reports 1/2 of wide coalesced streaming reads)",
               "traffic_bytes_per_frame": med * 1024 * 2, "kernels_per_frame": mode, "frames_sampled": len(steps)},
              open(out, "w"), indent=1)
    print(open(out).read())


def mfma(d, out, source):
    agg = defaultdict(lambda: [0, 0.0, 0.0])
    for e in per_dispatch(d):
        a = agg[e["name"][:90]]
        a[0] += 1; a[1] += e["c"].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0); a[2] += e["c"].get("GRBM_GUI_ACTIVE", 0.0)
    rows = [{"kernel": k, "launches": n, "mfma_busy_frac": (b / (g / 8 * 1024) if g else 0.0)} for k, (n, b, g) in agg.items()]
    rows.sort(key=lambda r: -r["mfma_busy_frac"])
    json.dump({"source": source, "note": "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs)",
               "kernels": rows[:40]}, open(out, "w"), indent=1)
    for r in rows[:12]:
        print(r)


def stats(d, out):
    files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    if not files:
        sys.exit(f"no *kernel_stats.csv under {d}")
    lines = open(files[0]).read().splitlines()
    open(out, "w").write("\n".join(lines[:45]) + "\n")
    print("\n".join(l[:150] for l in lines[:16]))


if __name__ == "__main__":
    mode, d, out = sys.argv[1:4]
    src = sys.argv[4] if len(sys.argv) > 4 else ""
    {"traffic": lambda: traffic(d, out, src), "mfma": lambda: mfma(d, out, src), "stats": lambda: stats(d, out)}[mode]()
