"""gpurun_out/<tag>/ (tools/profile_round2.sh) -> tracked files under profiles/<prefix>_*.
python tools/condense_round2.py r2a r2"""
import collections, csv, glob, json, os, shutil, subprocess, sys
tag, prefix = sys.argv[1], sys.argv[2]
src = os.path.join("gpurun_out", tag); dst = "profiles"
def newest(pattern):
    """gpurun merges every run into the same directory: take the files of the last one."""
    files = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    if not files: raise SystemExit("nothing matches " + pattern)
    return files[-1]
def line(path):
    return json.loads([l for l in open(path) if l.startswith("{")][-1])
for name in ("bench_driver_args", "bench_default"):
    json.dump(line(os.path.join(src, name + ".json")), open(os.path.join(dst, f"{prefix}_{name}.json"), "w"), indent=1)
notes = {"driver_args": "python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --no-4k-leg (the driver's arguments; headline leg)",
         "4k": "python3 bench.py --workload cornell_4k_16spp_per_step --steps 64 --warmup 2 --cpu-seconds 0 --no-4k-leg (the 4K / 1024 spp leg as a run of its own)",
         "default": "python3 bench.py --cpu-seconds 0 --no-4k-leg (1024 steps)",
         "serial": "python3 bench.py --frames-in-flight 1 --steps 128 --warmup 8 --cpu-seconds 0 --no-4k-leg"}
for t, note in notes.items():
    tr = newest(os.path.join(src, "trace_" + t, "**", "*kernel_trace.csv"))
    st = newest(os.path.join(src, "trace_" + t, "**", "*kernel_stats.csv"))
    shutil.copy(st, os.path.join(dst, f"{prefix}_kernel_stats_{t}.csv"))
    subprocess.run([sys.executable, "tools/trace_span.py", tr, os.path.join(src, f"trace_{t}.log"), os.path.join(dst, f"{prefix}_kernel_trace_span_{t}.json"),
                    f"rocprofv3 --kernel-trace --stats -- {note}; {src}/trace_{t}"], check=True, stdout=subprocess.DEVNULL)
def counters(sub, kernel_filter):
    agg = collections.defaultdict(float); disp = set()
    for f in [newest(os.path.join(src, sub, "**", "*counter_collection.csv"))]:
        for r in csv.DictReader(open(f)):
            if kernel_filter in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
    n = max(len(disp), 1)
    return {k: v / n for k, v in agg.items()}, len(disp)
plain = "pt_path_kernel<false, false"
out = {"source": f"tools/profile_round2.sh {tag}: rocprofv3 --pmc, one pass per counter group, bench.py --frames-in-flight 1 --steps 64 --no-4k-leg (720p, 1 spp per launch, serial launches with the display fused)",
       "kernel": plain, "per_launch": {}}
for sub in ("pmc_valu", "pmc_wait", "pmc_fetch", "pmc_write"):
    c, n = counters(sub, plain)
    out["per_launch"][sub] = dict(c, dispatches=n)
v = out["per_launch"]["pmc_valu"]
out["derived"] = {"valu_thread_utilisation": v["SQ_THREAD_CYCLES_VALU"] / (64.0 * v["SQ_INSTS_VALU"]),
                  "valu_insts_per_launch": v["SQ_INSTS_VALU"], "salu_insts_per_launch": v["SQ_INSTS_SALU"],
                  "lds_bank_conflict_cycles": out["per_launch"]["pmc_wait"].get("SQ_LDS_BANK_CONFLICT", 0.0)}
# MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half the bytes
# of 16-B-per-lane reads (the accumulator prefetch is one) -> doubled; WRITE_SIZE is exact for 16-B stores.
fetch = out["per_launch"]["pmc_fetch"]["FETCH_SIZE"] * 1024.0 * 2.0
write = out["per_launch"]["pmc_write"]["WRITE_SIZE"] * 1024.0
out["cornell_720p_progressive_1spp"] = {"hbm_bytes_per_launch": round(fetch + write), "fetch_bytes_corrected": round(fetch), "write_bytes": round(write),
                                         "algorithmic_bytes_per_launch": 1280 * 720 * (16 + 16 + 4),
                                         "note": "serial launch with the fused display: 16 B read + 16 B written + 4 B display per pixel.  An overlapped launch writes 16 B and its fold pass moves 16 + 16 + 16 + 4 B: 68 B per pixel and frame"}
json.dump(out, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
json.dump(out, open(os.path.join(dst, f"{prefix}_pmc_720p_serial.json"), "w"), indent=1)
print(json.dumps(out["derived"], indent=1)); print(out["cornell_720p_progressive_1spp"])
for t in notes:
    d = json.load(open(os.path.join(dst, f"{prefix}_kernel_trace_span_{t}.json")))
    print(t, round(d["ns_per_launch_incl_last_fold"]), d["bench_line_of_the_same_run"]["kernel_ms"], round(d["frac_from_this_trace"], 4), d["bench_line_of_the_same_run"]["frac"], round(d["frac_line_over_trace"], 4))
