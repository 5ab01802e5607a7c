"""Condense rocprofv3 output directories into a small text summary (committed under profiles/)."""
import csv, glob, os, sys, collections
root = sys.argv[1]

def find(pattern):
    return sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))

cmd = sys.argv[2] if len(sys.argv) > 2 else "python bench.py --steps 8 --warmup 2 --no-cpu-baseline"
print(f"== rocprofv3 --kernel-trace --stats of: {cmd}")
print("   (the GPU contexts of bench.py — six by default — run side by side in the timed region, so per-launch durations there measure sharing of the chip; the")
print("    'solo' table below keeps only launches that no launch of another queue overlaps = bench.py's single-context leg, the")
print("    region its roofline numbers are measured on)")
for f in find("*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    print(f"-- {os.path.relpath(f, root)}")
    print(f"{'kernel':70s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
    for r in rows[:25]:
        name = r.get("Name", "")[:70]
        print(f"{name:70s} {r.get('Calls',''):>6s} {float(r.get('TotalDurationNs',0))/1e6:10.3f} {float(r.get('AverageNs',0))/1e3:10.2f} "
              f"{float(r.get('MinNs',0))/1e3:9.2f} {float(r.get('MaxNs',0))/1e3:9.2f} {r.get('Percentage',''):>6s}")
# per-launch durations of the scatter kernel on full-size passes from the raw trace
for f in find("*kernel_trace.csv"):
    if "pmc" in f:
        continue
    rows = list(csv.DictReader(open(f)))
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "rs_onesweep_kernel<true" in r["Kernel_Name"] or "rs_scatter_kernel<true" in r["Kernel_Name"] or "rs_scatter_wc_kernel<true" in r["Kernel_Name"] or "rs_scatter_tiled_kernel<true" in r["Kernel_Name"]]
    if d:
        d.sort(reverse=True)
        big = [x for x in d if x > 300]
        print(f"-- {os.path.relpath(f, root)}: (key, value) digit-pass launches={len(d)}; launches > 300 us: {len(big)} avg {sum(big)/max(len(big),1):.1f} us")
        if big:
            print(f"   => full 64 Mi-record pass: 24 B x 67108864 / {sum(big)/len(big):.1f} us = {24*67108864/(sum(big)/len(big))/1e3:.0f} GB/s")
# launches not overlapped by any launch of another queue: per-kernel medians (what one block costs on an otherwise idle GPU)
import statistics
for f in find("*kernel_trace.csv"):
    if "pmc" in f:
        continue
    rows = list(csv.DictReader(open(f)))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"].split("(")[0][:64], int(r["Grid_Size_X"])) for r in rows)
    solo = collections.defaultdict(list)
    for i, (s0, e0, q0, nm, g) in enumerate(ev):
        if any(q1 != q0 and s1 < e0 and e1 > s0 for (s1, e1, q1, _, _) in ev[max(0, i - 48):i + 48]):
            continue
        solo[nm].append((e0 - s0) / 1e3)
    print(f"-- {os.path.relpath(f, root)}: launches no other queue overlaps")
    print(f"{'kernel':66s} {'n':>4s} {'median_us':>10s} {'mean_us':>10s} {'min_us':>9s}")
    for nm, v in sorted(solo.items(), key=lambda kv: -sum(kv[1]))[:40]:
        print(f"{nm:66s} {len(v):4d} {statistics.median(v):10.1f} {sum(v)/len(v):10.1f} {min(v):9.1f}")
    full = [x for nm, v in solo.items() if nm.startswith(("void rs_onesweep_kernel<true", "void rs_scatter_kernel<true, 1024", "void rs_scatter_wc_kernel<true", "void rs_scatter_tiled_kernel<true")) for x in v if x > 200]
    if full:
        print(f"   => solo full-size (key, value) digit passes (rs_onesweep / rs_scatter_tiled / rs_scatter_wc / rs_scatter 1024 x 8): {len(full)} launches, mean {sum(full)/len(full):.1f} us -> 24 B x 67108864 / mean = {24*67108864/(sum(full)/len(full))/1e3:.0f} GB/s")
    ha = [x for nm, v in solo.items() if nm.startswith("rs_hist_all_kernel") for x in v if x > 100]
    if full and ha:
        sort_us = 8 * sum(full) / len(full) + sum(ha) / len(ha)
        print(f"   => solo first sort = rs_hist_all ({sum(ha)/len(ha):.1f} us) + 8 digit passes = {sort_us:.1f} us for B_sort = 8 B x m + 8 x 24 B x m = {(8 + 8 * 24) * 67108864 / 1e9:.2f} GB -> {(8 + 8 * 24) * 67108864 / sort_us / 1e3:.0f} GB/s")
print()
print("== PMC passes (one whole 64 MiB block: sorter, QLFC front end, device coder; counters per dispatch, summed per kernel; FETCH_SIZE/WRITE_SIZE in KiB units as reported)")
for tag, cname in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    for f in find("*counter_collection.csv"):
        if tag not in f:
            continue
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != cname:
                continue
            k = r["Kernel_Name"].split("(")[0][:60]
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
        print(f"-- {cname} ({os.path.relpath(f, root)})")
        for k, (cnt, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
            print(f"{k:62s} dispatches {cnt:4d}  sum {v:16.0f}  per-dispatch {v/cnt:14.0f}")

# machine-readable traffic for bench.py's roofline.traffic
import json
def per_launch(tag, cname, kname):
    for f in find("*counter_collection.csv"):
        if tag not in f: continue
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r.get("Counter_Name") == cname and kname in r["Kernel_Name"]]
        return vals
    return []
fe = per_launch("pmc_fetch", "FETCH_SIZE", "rs_onesweep_kernel<true") or per_launch("pmc_fetch", "FETCH_SIZE", "rs_scatter_tiled_kernel<true") or per_launch("pmc_fetch", "FETCH_SIZE", "rs_scatter_wc_kernel<true") or per_launch("pmc_fetch", "FETCH_SIZE", "rs_scatter_kernel<true")
wr = per_launch("pmc_write", "WRITE_SIZE", "rs_onesweep_kernel<true") or per_launch("pmc_write", "WRITE_SIZE", "rs_scatter_tiled_kernel<true") or per_launch("pmc_write", "WRITE_SIZE", "rs_scatter_wc_kernel<true") or per_launch("pmc_write", "WRITE_SIZE", "rs_scatter_kernel<true")
if fe and wr:
    nfull = 8
    fetch_kib = sum(fe[:nfull]) / nfull; write_kib = sum(wr[:nfull]) / nfull
    out = {"records": 67108864, "kernel": "rs_onesweep_kernel<true>" if per_launch("pmc_fetch", "FETCH_SIZE", "rs_onesweep_kernel<true") else "rs_scatter",
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), the 8 digit passes of the first sort of one 64 MiB block",
           "rs_scatter_pairs": {"fetch_size_kib_reported": fetch_kib, "write_size_kib_reported": write_kib,
                                "traffic_bytes_per_launch": int((2 * fetch_kib + write_kib) * 1024),
                                "algorithmic_bytes_per_launch": 24 * 67108864,
                                "note": "FETCH_SIZE doubled per the gfx950 rule for wide coalesced loads; this over-corrects the 4-B/lane value loads (true read = 768 MiB)"}}
    json.dump(out, open(os.path.join(root, "pmc_traffic.json"), "w"), indent=1)
    print("pmc_traffic.json:", out["rs_scatter_pairs"])
