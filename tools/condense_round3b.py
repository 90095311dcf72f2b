"""gpurun_out/<tag>/ (tools/profile_round3b.sh) -> tracked files under profiles/<prefix>_*.
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
         "serial": "PT_AMD_RENDER_AHEAD=0 python3 bench.py --frames-in-flight 1 --steps 128 --warmup 8 --cpu-seconds 0 --no-4k-leg (one launch per frame, strictly serial)",
         "ahead": "python3 bench.py --frames-in-flight 1 --steps 128 --warmup 8 --cpu-seconds 0 --no-4k-leg (one frame at a time as the library runs it by default: the next frame rendered ahead)"}
for t, note in notes.items():
    if not glob.glob(os.path.join(src, "trace_" + t, "**", "*kernel_trace.csv"), recursive=True): continue
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
out = {"source": f"tools/profile_round3b.sh {tag}: rocprofv3 --pmc, one pass per counter group, bench.py --frames-in-flight 1 --steps 64 --no-4k-leg (720p, 1 spp per launch, serial launches with the display fused)",
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
out["serial_one_frame_launch"] = {"hbm_bytes_per_launch": round(fetch + write), "fetch_bytes_corrected": round(fetch), "write_bytes": round(write),
                                  "algorithmic_bytes_per_launch": 1280 * 720 * (16 + 16 + 4),
                                  "note": "serial launch with the fused display: 16 B read + 16 B written + 4 B display per pixel"}
json.dump(out, open(os.path.join(dst, f"{prefix}_pmc_720p_serial.json"), "w"), indent=1)

# ---- the driver's command: one launch of 20 frames + one fold pass
def nth_dispatch(sub, kernel_filter, n):
    """Counters of the n-th dispatch (in dispatch order) of the kernels whose name contains kernel_filter."""
    rows = [r for r in csv.DictReader(open(newest(os.path.join(src, sub, "**", "*counter_collection.csv")))) if kernel_filter in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    pick = ids[n]
    agg = collections.defaultdict(float)
    for r in rows:
        if int(r["Dispatch_Id"]) == pick: agg[r["Counter_Name"]] += float(r["Counter_Value"])
    return dict(agg)
_r = line(os.path.join(src, "bench_driver_args.json"))["roofline"]
frames = _r.get("steps_per_launch", _r.get("frames_per_launch"))
multi = "pt_path_kernel<false, false, false>"   # the build overlapped launches use; dispatches: set-up, warm-up, the timed launch
mf = {"source": f"tools/profile_round3b.sh {tag}: rocprofv3 --pmc, one pass per counter group, bench.py --gpus 1 --steps 20 --warmup 5 --no-4k-leg: "
                "the third dispatch of the path kernel is the launch that holds the 20 timed frames, the third fold pass the one behind it",
      "frames_per_launch": frames, "path_launch": {}, "fold_pass": {}}
for sub in ("pmcf_valu", "pmcf_fetch", "pmcf_write"):
    mf["path_launch"][sub] = nth_dispatch(sub, multi, 2)
    mf["fold_pass"][sub] = nth_dispatch(sub, "pt_fold_kernel", 2)
def traffic(d):
    f = d["pmcf_fetch"]["FETCH_SIZE"] * 1024.0 * 2.0; w = d["pmcf_write"]["WRITE_SIZE"] * 1024.0
    return round(f + w), round(f), round(w)
npx = 1280 * 720
t, f_, w_ = traffic(mf["path_launch"])
v2 = mf["path_launch"]["pmcf_valu"]
mf["path_launch"]["derived"] = {"hbm_bytes": t, "fetch_bytes_corrected": f_, "write_bytes": w_, "hbm_bytes_per_frame": round(t / frames),
                                "algorithmic_bytes": round(npx * 16 * frames), "valu_thread_utilisation": v2["SQ_THREAD_CYCLES_VALU"] / (64.0 * v2["SQ_INSTS_VALU"]),
                                "valu_insts": v2["SQ_INSTS_VALU"], "note": "one 16-B store of (sum r, g, b, count) per pixel and frame; nothing is read"}
t2, f2, w2 = traffic(mf["fold_pass"])
mf["fold_pass"]["derived"] = {"hbm_bytes": t2, "fetch_bytes_corrected": f2, "write_bytes": w2, "algorithmic_bytes": round(npx * (16 * frames + 16 + 16 + 4 * frames)),
                              "note": "per pixel: 16 B read per frame, 16 B read + 16 B written for the accumulation buffer, 4 B written per frame's display"}
json.dump(mf, open(os.path.join(dst, f"{prefix}_pmc_720p_frames.json"), "w"), indent=1)
summary = {"source": f"profiles/{prefix}_pmc_720p_frames.json, profiles/{prefix}_pmc_720p_serial.json",
           "cornell_720p_progressive_1spp": {"hbm_bytes_per_frame": mf["path_launch"]["derived"]["hbm_bytes_per_frame"], "frames_per_launch_measured": frames,
                                             "hbm_bytes_per_launch": t, "algorithmic_bytes_per_frame": npx * 16,
                                             "fold_pass_hbm_bytes": t2, "serial_one_frame_launch_hbm_bytes": out["serial_one_frame_launch"]["hbm_bytes_per_launch"]}}
json.dump(summary, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
print(json.dumps(out["derived"], indent=1)); print(json.dumps(summary, indent=1)); print(mf["path_launch"]["derived"])
for t in notes:
    d = json.load(open(os.path.join(dst, f"{prefix}_kernel_trace_span_{t}.json")))
    print(t, round(d["ns_per_launch_incl_last_fold"]), d["bench_line_of_the_same_run"]["kernel_ms"], round(d["frac_from_this_trace"], 4), d["bench_line_of_the_same_run"]["frac"], round(d["frac_line_over_trace"], 4))
