"""gpurun_out/direct_pmc (tools/direct_pmc.sh: two rocprofv3 --pmc passes over tools/direct_run.py) -> profiles/<tag>_pmc_direct.json:
per launch of pt_direct_kernel (mean over the launches of the newest CSV of each pass), per sample of 64 pixels, and the
issue cycles those instructions cost at the rates tools/ubench/f64_rates.hip measured (profiles/r3_f64_rates.json)."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r3"
note = sys.argv[2] if len(sys.argv) > 2 else ""
counters = {}
for sub in ("valu", "wait"):
    files = glob.glob(os.path.join(ROOT, "gpurun_out", "direct_pmc", sub, "*", "*_counter_collection.csv"))
    f = max(files, key=os.path.getmtime)
    per = {}
    for row in csv.DictReader(open(f)):
        if "pt_direct_kernel" not in row["Kernel_Name"]:
            continue
        per.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
        per[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    for name, by_dispatch in per.items():
        counters[name] = round(sum(by_dispatch.values()) / len(by_dispatch))
W, H, SPP = 1280, 720, 64
iters = W * H * SPP // 64
c = counters
per_it = {"valu": c["SQ_INSTS_VALU"] / iters, "salu": c["SQ_INSTS_SALU"] / iters, "smem": c["SQ_INSTS_SMEM"] / iters,
          "f64_add": c["SQ_INSTS_VALU_ADD_F64"] / iters, "f64_mul": c["SQ_INSTS_VALU_MUL_F64"] / iters,
          "f64_fma": c["SQ_INSTS_VALU_FMA_F64"] / iters, "f64_trans": c["SQ_INSTS_VALU_TRANS_F64"] / iters}
per_it["valu_not_f64"] = per_it["valu"] - per_it["f64_add"] - per_it["f64_mul"] - per_it["f64_fma"] - per_it["f64_trans"]
rates = json.load(open(os.path.join(ROOT, "profiles", "r3_f64_rates.json")))
cyc = rates.get("cycles_per_wave_instruction", rates)
def rate(*names, default):
    for n in names:
        if n in cyc: return float(cyc[n])
    return default
priced = (per_it["f64_fma"] * rate("v_fma_f64", "fma", default=5.0) + (per_it["f64_add"] + per_it["f64_mul"]) * rate("v_add_f64", "add", default=4.5) +
          per_it["f64_trans"] * rate("v_rcp_f64", "rcp", default=16.3) + per_it["valu_not_f64"] * 4.0)
simds = 256 * 4
gpu_cycles = c["GRBM_GUI_ACTIVE"] / 8.0   # (the counter is summed over the 8 XCDs)
cycles_per_it = gpu_cycles * simds / iters
out = {"workload": "cpu_default 1280x720, `direct`, FP64, 64 spp per launch (BASELINE config 1); tools/direct_pmc.sh, per launch (mean over the launches of the run)" + (f"; {note}" if note else ""),
       "kernel": "pt_direct_kernel", "wave_iterations_per_launch": iters, "note_iter": "one iteration = one sample of 64 pixels (a wave's trip through the sample loop)",
       "counters": dict(sorted(counters.items())), "per_iteration": {k: round(v, 1) for k, v in per_it.items()},
       "valu_thread_utilisation": round(c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64.0), 3) if c.get("SQ_ACTIVE_INST_VALU") else None,
       "gpu_cycles_per_launch": round(gpu_cycles), "cycles_per_iteration_and_simd": round(cycles_per_it),
       "issue_cycles_priced_with_r3_f64_rates": round(priced), "valu_pipe_busy": round(priced / cycles_per_it, 2),
       "note_pricing": "fma 5.0, add / mul 4.5, rcp / rsq 16.3 cycles per wave64 instruction as measured; 4.0 assumed for the VALU instructions that are not FP64 arithmetic"}
path = os.path.join(ROOT, "profiles", f"{tag}_pmc_direct.json")
json.dump(out, open(path, "w"), indent=1)
print(path, json.dumps(out["per_iteration"]), out["valu_thread_utilisation"], out["cycles_per_iteration_and_simd"], out["issue_cycles_priced_with_r3_f64_rates"], out["valu_pipe_busy"])
