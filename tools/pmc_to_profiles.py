"""Condense gpurun_out/<tag>/ (tools/profile_round.sh) into tracked files under profiles/.
python tools/pmc_to_profiles.py r1_v3"""
import collections, csv, glob, json, os, shutil, sys
tag = sys.argv[1]
src = os.path.join("gpurun_out", tag); dst = "profiles"
def counters(sub, kernel_filter):
    agg = collections.defaultdict(float); disp = set()
    for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel_filter in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
    n = max(len(disp), 1)
    return {k: v / n for k, v in agg.items()}, len(disp)
plain = "pt_path_kernel<false, false"
out = {"source": f"tools/profile_round.sh {tag}: rocprofv3 --pmc, one pass per counter group, bench.py --frames-in-flight 1 --steps 64 (720p, 1 spp per launch)",
       "kernel": plain, "per_launch": {}}
for sub in ("pmc_valu", "pmc_wait", "pmc_fetch", "pmc_write"):
    c, n = counters(sub, plain)
    out["per_launch"][sub] = dict(c, dispatches=n)
v = out["per_launch"]["pmc_valu"]
out["derived"] = {
    "valu_thread_utilisation": v["SQ_THREAD_CYCLES_VALU"] / (64.0 * v["SQ_INSTS_VALU"]),
    "valu_insts_per_launch": v["SQ_INSTS_VALU"], "salu_insts_per_launch": v["SQ_INSTS_SALU"],
    "lds_bank_conflict_cycles": out["per_launch"]["pmc_wait"].get("SQ_LDS_BANK_CONFLICT", 0.0),
}
# MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half the bytes
# of 16-B-per-lane reads (the accumulator prefetch is one) -> doubled; WRITE_SIZE is exact for 16-B stores.
fetch = out["per_launch"]["pmc_fetch"]["FETCH_SIZE"] * 1024.0 * 2.0
write = out["per_launch"]["pmc_write"]["WRITE_SIZE"] * 1024.0
out["cornell_720p_progressive_1spp"] = {"hbm_bytes_per_launch": round(fetch + write), "fetch_bytes_corrected": round(fetch), "write_bytes": round(write),
                                         "algorithmic_bytes_per_launch": 1280 * 720 * (16 + 16 + 4)}
json.dump(out, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
for name in ("bench.json", "bench_serial.json", "bench_4k.json"):
    shutil.copy(os.path.join(src, name), os.path.join(dst, f"{tag}_{name}"))
for sub, name in (("trace_default", "kernel_stats_flight8"), ("trace_serial", "kernel_stats_serial")):
    f = glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True)[0]
    shutil.copy(f, os.path.join(dst, f"{tag}_{name}.csv"))
print(json.dumps(out["derived"], indent=1)); print(out["cornell_720p_progressive_1spp"])
