"""Copies the judged summaries of a tools/profile_bench.sh run from gpurun_out/prof_<tag>/
into profiles/<name>/ : kernel_stats.csv, domain_stats.csv, pmc_summary.json, and writes
profiles/pmc_traffic.json (HBM bytes per launch of the dominant kernel, for bench.py)."""
import collections, csv, glob, json, os, shutil, sys

tag, name = sys.argv[1], sys.argv[2]
workload = sys.argv[3] if len(sys.argv) > 3 else "c1"   # which whole-step figure of profiles/pmc_traffic.json this run supplies
src = "gpurun_out/prof_%s" % tag
dst = "profiles/%s" % name
os.makedirs(dst, exist_ok=True)
for kind in ("kernel_stats", "domain_stats"):
    f = glob.glob("%s/trace/*/*%s.csv" % (src, kind))
    if f:
        shutil.copy(f[0], "%s/%s.csv" % (dst, kind))
out = {}
for sub in ("pmc_fetch", "pmc_write"):
    f = glob.glob("%s/%s/*/*counter_collection.csv" % (src, sub))
    if not f:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        for c, x in v.items():
            out.setdefault(k, {})[c] = {"launches": len(x), "avg": sum(x) / len(x)}
json.dump(out, open("%s/pmc_summary.json" % dst, "w"), indent=1, sort_keys=True)
# per-launch HBM traffic of each kernel, corrected as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half of a wide coalesced read.
traffic = {}
for k, v in out.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        traffic[k] = {"fetch_bytes_raw": v["FETCH_SIZE"]["avg"] * 1024, "fetch_bytes_x2": v["FETCH_SIZE"]["avg"] * 2048,
                      "write_bytes": v["WRITE_SIZE"]["avg"] * 1024}
stats = {}
f = "%s/kernel_stats.csv" % dst
if os.path.exists(f):
    for r in csv.DictReader(open(f)):
        stats[r["Name"].split("(")[0]] = float(r["AverageNs"])
json.dump({"per_kernel": traffic, "avg_ns": stats}, open("%s/traffic.json" % dst, "w"), indent=1, sort_keys=True)
# what bench.py's roofline.traffic reads: HBM bytes per launch of each main kernel
per = {}
for k, t in traffic.items():
    n = k.replace("void ", "")
    if n.startswith(("k_partition", "k_window", "k_build_direct", "k_index_text")):
        per[n] = t["fetch_bytes_x2"] + t["write_bytes"]
# HBM bytes of ONE step (bench.py's roofline.traffic): every kernel of the build, launches per step from the trace
whole = None
if "k_index_text" in out and "FETCH_SIZE" in out["k_index_text"]:
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 4   # tools/profile_bench.sh runs --steps 3 --warmup 1
    whole = 0.0
    for k, t in traffic.items():
        n = k.replace("void ", "")
        if n.startswith("k_") and not n.startswith(("k_random_access", "k_lds_rate")):
            whole += (t["fetch_bytes_x2"] + t["write_bytes"]) * out[k]["FETCH_SIZE"]["launches"] / steps
if per:
    cur = {}
    if os.path.exists("profiles/pmc_traffic.json"):
        cur = json.load(open("profiles/pmc_traffic.json"))
    key = "whole_step_hbm_bytes" if workload == "c1" else "whole_step_hbm_bytes_" + workload
    cur[key] = whole
    cur.setdefault("sources", {})[key] = dst + "/traffic.json"
    if workload == "c1":
        old = cur.get("per_kernel_hbm_bytes_per_launch", {})
        old.update(per)
        cur["per_kernel_hbm_bytes_per_launch"] = old
    cur["source"] = ("(TCC FETCH_SIZE*2 + WRITE_SIZE) KiB per launch summed over the kernels of one bench step, separate --pmc passes, "
                     "gfx950 FETCH_SIZE x2 correction (MI355X_MICROARCH.md HBM section); whole_step_hbm_bytes = configs[1], "
                     "whole_step_hbm_bytes_c2 = configs[2] on one GPU; see `sources` for the run behind each")
    json.dump(cur, open("profiles/pmc_traffic.json", "w"), indent=1, sort_keys=True)
for k in sorted(stats, key=lambda k: -stats[k])[:8]:
    t = traffic.get(k, {})
    print("%-45s %9.1f us  fetch(x2) %8.1f MB  write %8.1f MB" % (k[:45], stats[k] / 1e3, t.get("fetch_bytes_x2", 0) / 1e6, t.get("write_bytes", 0) / 1e6))
