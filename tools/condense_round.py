"""Everything tools/profile_round.sh + the extra bench runs + tools/pmc_4k.sh left under gpurun_out/ -> profiles/.
python tools/condense_round.py r1_v5"""
import csv, glob, json, os, re, shutil, subprocess, sys
tag = sys.argv[1]
src = os.path.join("gpurun_out", tag)
subprocess.run([sys.executable, "tools/pmc_to_profiles.py", tag], check=True, stdout=subprocess.DEVNULL)
for name in ("glass", "random4096", "mis", "flight3", "flight4"):
    f = os.path.join(src, f"bench_{name}.json")
    if os.path.exists(f): shutil.copy(f, f"profiles/{tag}_bench_{name}.json")
# amortised time per launch from the kernel trace of the default bench run
f = glob.glob(os.path.join(src, "trace_default", "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "pt_path_kernel<false, false" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
st = [int(r["Start_Timestamp"]) for r in rows]; en = [int(r["End_Timestamp"]) for r in rows]
log = open(os.path.join(src, "trace_default.log")).read()
ev = {k: float(re.search(r'"%s": ([0-9.]+)' % k, log).group(1)) for k in ("kernel_ms", "kernel_ms_serial", "value", "steps", "warmup")}
W, K = int(ev["warmup"]), int(ev["steps"])
timed = range(W, W + K); span = max(en[i] for i in timed) - min(st[i] for i in timed); busy = sum(en[i] - st[i] for i in timed)
ser = range(W + K, len(rows)); span_s = max(en[i] for i in ser) - min(st[i] for i in ser)
json.dump({"source": f"rocprofv3 --kernel-trace of `python3 bench.py --cpu-seconds 0` ({src}/trace_default): pt_path_kernel<false,false> dispatches in start order; {W} warm-up, {K} timed with frames in flight, {len(ser)} strictly serial",
           "timed_launches": K, "first_start_to_last_end_ns": span, "ns_per_launch_amortised": span / K, "mean_dispatch_duration_ns": busy / K,
           "mean_launches_in_flight": busy / span, "serial_launches": len(ser), "serial_ns_per_launch": span_s / len(ser),
           "bench_events_in_the_same_run": {"kernel_ms": ev["kernel_ms"], "kernel_ms_serial": ev["kernel_ms_serial"], "value_msamples_per_s": ev["value"]}},
          open(f"profiles/{tag}_kernel_trace_span.json", "w"), indent=1)
# 4K PMC passes (tools/pmc_4k.sh)
if os.path.exists("gpurun_out/pmc_4k_a.json"):
    out = {"source": "tools/pmc_4k.sh: rocprofv3 --pmc, three passes, tools/pt_headless --width 3840 --height 2160 --spp 32 --spp-per-launch 16 (cuda_default, 2 launches of pt_path_kernel<false,false>); sums over both launches"}
    for p in "abc": out["pass_" + p] = json.load(open(f"gpurun_out/pmc_4k_{p}.json"))
    a, b, c = out["pass_a"], out["pass_b"], out["pass_c"]
    rays = 1715069583.36  # event counters of the same 2 x 16 spp (bench.py --workload cornell_4k_16spp_per_step: rays_per_sample x samples)
    out["derived"] = {"valu_per_full_64_ray_trip": a["SQ_INSTS_VALU"] / (rays / 64.0), "salu_per_trip": a["SQ_INSTS_SALU"] / (rays / 64.0),
                      "valu_thread_utilisation": a["SQ_THREAD_CYCLES_VALU"] / (64.0 * a["SQ_INSTS_VALU"]), "rays": rays,
                      "waves_issuing_frac": b["SQ_ACTIVE_INST_ANY"] / a["SQ_WAVE_CYCLES"], "waves_waitcnt_frac": b["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"],
                      "waves_waiting_for_issue_frac": b["SQ_WAIT_INST_ANY"] / a["SQ_WAVE_CYCLES"],
                      "valu_pipe_busy": a["SQ_INSTS_VALU"] / 1024.0 * 2.0 / (b["GRBM_GUI_ACTIVE"] / 8.0),
                      "mix": {k[14:]: round(c[k] / a["SQ_INSTS_VALU"], 4) for k in c if k.startswith("SQ_INSTS_VALU_")}}
    json.dump(out, open(f"profiles/{tag}_pmc_4k.json", "w"), indent=1)
for f in sorted(glob.glob(f"profiles/{tag}_bench*.json")):
    d = json.load(open(f)); r = d["roofline"]
    print(f.split("bench")[-1], d["value"], d["mrays_per_s"], r["frac"], r["achieved"], d["ms_per_step"], r["kernel_ms_serial"], r["frac_serial"], d["config"]["frames_in_flight"])
print(json.load(open(f"profiles/{tag}_kernel_trace_span.json")))
