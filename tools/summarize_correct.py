"""Copies the judged summaries of a tools/profile_correct.sh run from gpurun_out/prof_ec_<tag>/ into
profiles/<name>/: bench.json, kernel_stats.csv, pmc_summary.json (per kernel: average counter values
per launch; FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them)."""
import collections, csv, glob, json, os, shutil, sys

tag, name = sys.argv[1], sys.argv[2]
src = "gpurun_out/prof_ec_%s" % tag
dst = "profiles/%s" % name
os.makedirs(dst, exist_ok=True)
shutil.copy("%s/bench.json" % src, "%s/bench.json.log" % dst)
for kind in ("kernel_stats", "domain_stats"):
    f = glob.glob("%s/trace/*/*%s.csv" % (src, kind))
    if f:
        shutil.copy(f[0], "%s/%s.csv" % (dst, kind))
out = {}
for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
    for f in glob.glob("%s/%s/*/*counter_collection.csv" % (src, sub)):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            if "k_correct" in k or "k_ec_" in k or "k_random" in k or "k_index" in k:
                for c, x in v.items():
                    out.setdefault(k, {})[c] = {"launches": len(x), "avg": sum(x) / len(x)}
json.dump(out, open("%s/pmc_summary.json" % dst, "w"), indent=1, sort_keys=True)
for r in list(csv.DictReader(open("%s/kernel_stats.csv" % dst)))[:8]:
    print("%-50s calls %3s avg %10.1f us" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3))
for k, v in out.items():
    if "k_correct" in k:
        print(k, {c: round(x["avg"] / 1e6, 1) for c, x in v.items()})
