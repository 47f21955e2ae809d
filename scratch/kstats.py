#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table from the gfx950 .s file(s) hipcc leaves with -save-temps=obj.
usage: scratch/kstats.py file.s [name-filter]"""
import re, sys, subprocess
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cur = None
rows = []
for line in txt.splitlines():
    m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
    if m:
        cur = m.group(1)
        continue
    m = re.match(r"^; (NumVgprs|NumAgprs|TotalNumVgprs|ScratchSize|Occupancy|LDSByteSize|NumSgprs): (\d+)", line)
    if m and cur:
        if not rows or rows[-1][0] != cur:
            rows.append((cur, {}))
        rows[-1][1][m.group(1)] = int(m.group(2))
for name, d in rows:
    dem = subprocess.run(["/usr/bin/c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
    if flt and flt not in dem:
        continue
    print(f"{dem:70s} vgpr {d.get('NumVgprs'):4d} agpr {d.get('NumAgprs'):3d} total {d.get('TotalNumVgprs'):4d} scratch {d.get('ScratchSize'):5d} lds {d.get('LDSByteSize'):6d} occ {d.get('Occupancy')}")
