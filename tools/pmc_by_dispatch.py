#!/usr/bin/env python3
"""Per-dispatch counter values of one kernel from a rocprofv3 --pmc run (csv): usage: pmc_by_dispatch.py <dir> <kernel substring>"""
import csv, glob, sys, collections
d, pat = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
rows = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if pat not in r["Kernel_Name"]:
        continue
    rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
names = sorted({n for v in rows.values() for n in v})
print("dispatch", *names)
for k, v in rows.items():
    print(k, *["%.4g" % v.get(n, float("nan")) for n in names])
