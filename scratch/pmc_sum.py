"""Summarise rocprofv3 --pmc csv output: mean counter value per launch for the message kernels."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "message" not in k and "probe" not in k and "mlp2" not in k and "node_block" not in k:
                continue
            k = k.split("(")[0][-36:]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(d.split("/")[-1], k, {c: f"{sum(v)/len(v):.4g}" for c, v in sorted(cs.items())}, "launches", len(next(iter(cs.values()))))
