"""Summarise the two rocprofv3 --pmc passes of profiles/collect_traffic.sh into
profiles/<tag>_traffic_pmc.csv (per kernel: launches, mean FETCH_SIZE, mean WRITE_SIZE, as reported, in KiB) and
profiles/traffic.json ({workload: {C-ABI entry point: HBM bytes per launch}}), which bench.py puts into
roofline.traffic.

Corrections (MI355X_MICROARCH.md, HBM section): on gfx950 FETCH_SIZE = TCC_EA0_RDREQ x 64 B tallies a 128-byte
request as 64 bytes for wide coalesced streaming reads, so it is DOUBLED; WRITE_SIZE is exact for 16-B-per-lane
stores.  The message kernels read dword-per-lane row gathers, an access width the guide marks as uncalibrated: both
the raw and the doubled figure are written to the csv, and traffic.json carries 2 x FETCH + WRITE.
"""
import csv, glob, json, os, sys, collections

tag, d_fetch, d_write = sys.argv[1:4]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def collect(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


fetch, write = collect(d_fetch, "FETCH_SIZE"), collect(d_write, "WRITE_SIZE")
rows = []
for k in sorted(set(fetch) | set(write)):
    f = sum(fetch.get(k, [0])) / max(1, len(fetch.get(k, [])))
    w = sum(write.get(k, [0])) / max(1, len(write.get(k, [])))
    rows.append((k, len(fetch.get(k, [])), f, w))
rows.sort(key=lambda r: -(r[2] * r[1]))
out_csv = os.path.join(ROOT, "profiles", f"{tag}_traffic_pmc.csv")
with open(out_csv, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "launches", "FETCH_SIZE_KiB_mean_raw", "FETCH_x2_MB", "WRITE_SIZE_KiB_mean", "HBM_MB_per_launch(2*F+W)"])
    for k, n, f, wr in rows:
        w.writerow([k[:120], n, f"{f:.1f}", f"{2 * f * 1024 / 1e6:.2f}", f"{wr:.1f}", f"{(2 * f + wr) * 1024 / 1e6:.2f}"])
print(open(out_csv).read()[:3000])

# kernel symbol -> C-ABI entry point names used by bench.py's KERNEL_TIMER
alias = {"k_message_fwd_sb": "xeq_message_fwd_sb", "k_message_bwd_sb": "xeq_message_bwd_sb",
         "k_message_fwd_wq": "xeq_message_fwd_wq", "k_message_bwd_wq": "xeq_message_bwd_wq"}
traffic = {}
for k, n, f, wr in rows:
    for sym, name in alias.items():
        if sym in k:
            # the wq kernels' first-block forms (second template argument true) are separate kernels with their own traffic
            first = sym.endswith("_wq") and ", true>" in k.split("(")[0]
            traffic[name + ("_first" if first else "")] = (2 * f + wr) * 1024
tfile = os.path.join(ROOT, "profiles", "traffic.json")
allw = json.load(open(tfile)) if os.path.exists(tfile) else {}
allw["qm9_1024"] = traffic
allw["_note"] = ("HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, rocprofv3 --pmc, separate passes, "
                 f"profiles/{tag}_traffic_pmc.csv; FETCH doubling per MI355X_MICROARCH.md (uncalibrated for dword gathers)")
json.dump(allw, open(tfile, "w"), indent=1)
print(json.dumps(allw, indent=1))
