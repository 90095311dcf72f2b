"""gpurun_out/pmc_r3_<config>_{a,b,c,counters}/ (tools/profile_round3.sh) -> profiles/r3_pmc_<config>.json:
counters of the path kernel summed over the run's launches, the figures derived from them, the event counters and the
path-length ("bounce") histogram of the same samples.   python tools/condense_round3.py"""
import collections, csv, glob, json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def pmc(d):
    agg = collections.defaultdict(float); disp = set(); names = set()
    # (gpurun merges a call's files into gpurun_out/ beside those of earlier calls: the newest CSV is this run's)
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1:]:
        for r in csv.DictReader(open(f)):
            if "pt_path_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"]); names.add(r["Kernel_Name"].split("(")[0])
    out = dict(agg); out["dispatches_seen"] = len(disp); out["kernels"] = sorted(names)
    return out
summary = {}
for cfg in ("glass", "random4096", "nee", "cornell720", "cornell4k"):
    base = os.path.join(ROOT, "gpurun_out", f"pmc_r3_{cfg}_")
    if not os.path.isdir(base + "a"): continue
    a, b, c = pmc(base + "a"), pmc(base + "b"), pmc(base + "c")
    lines = [json.loads(l) for l in open(base + "counters/run.log") if l.startswith("{")]
    run, cnt = lines[0], lines[1]
    k, hist = cnt["counters"], cnt["path_length_histogram"]
    rays = k["rays"]; trips = rays / 64.0
    n = sum(hist)
    cum = lambda lo: sum(hist[lo:]) / n
    mix = {m: c[f"SQ_INSTS_VALU_{m}"] / a["SQ_INSTS_VALU"] for m in ("FMA_F32", "ADD_F32", "MUL_F32", "TRANS_F32", "INT32", "CVT")}
    out = {"source": "tools/profile_round3.sh: rocprofv3 --pmc, three passes (a, b, c), tools/pt_headless " + open(base + "counters/args.txt").read().strip()
                     + "; sums over the run's launches of the path kernel; event counters and histogram from a fourth run of the same samples with --counters",
           "pass_a": a, "pass_b": b, "pass_c": c, "event_counters": k, "path_length_histogram": hist,
           "derived": {"valu_thread_utilisation": a["SQ_THREAD_CYCLES_VALU"] / (64.0 * a["SQ_INSTS_VALU"]),
                       "valu_per_full_64_ray_trip": a["SQ_INSTS_VALU"] / trips, "salu_per_trip": a["SQ_INSTS_SALU"] / trips,
                       "branches_per_trip": b["SQ_INSTS_BRANCH"] / trips, "lds_insts_per_trip": b["SQ_INSTS_LDS"] / trips,
                       "lds_bank_conflict_cycles": c["SQ_LDS_BANK_CONFLICT"],
                       "waves_issuing_frac": b["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], "waves_waitcnt_frac": b["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
                       "mix": {m: round(v, 4) for m, v in mix.items()},
                       "rays": rays, "rays_per_sample": rays / k["samples"], "path_rays_per_sample": sum(i * h for i, h in enumerate(hist)) / n,
                       "paths_with_at_least_8_rays": cum(8), "at_least_16": cum(16), "at_least_32": cum(32), "longest_path": max(i for i, h in enumerate(hist) if h),
                       "msamples_per_s_of_the_counters_run_unprofiled": run["msamples_per_s"]}}
    json.dump(out, open(os.path.join(ROOT, "profiles", f"r3_pmc_{cfg}.json"), "w"), indent=1)
    d = out["derived"]
    summary[cfg] = {kk: (round(v, 4) if isinstance(v, float) else v) for kk, v in d.items() if kk != "mix"}
    print(cfg, json.dumps(summary[cfg]))
