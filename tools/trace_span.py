"""rocprofv3 --kernel-trace CSV of a bench.py run -> the time per launch the bench line's roofline is quoted on.

bench.py times its K frames with HIP events on the launch stream.  With frames in flight the library packs them into
L launches of the path kernel (up to 32 frames each; the bench line says how many: roofline.launches) which overlap,
so rocprofv3's per-dispatch duration (start to end of ONE dispatch, other dispatches running beside it) is up to
`frames in flight` times the time per launch.  The like-for-like figure from a trace is first start to last end of
the L timed dispatches / L.
Dispatch order of pt_path_kernel in a one-leg bench run (--no-4k-leg): set-up, warm-up, the L timed launches, then (if
frames in flight > 1) the two one-frame-at-a-time probes of m = 4 + n frames each (4 untimed, n = min(K - 4, 128) timed):
m strictly serial launches, then the same frames with render-ahead (2 launches on the context's stream, m - 2 launched
ahead and used, 1 launched ahead and left: m + 1), then the instrumented launches (template argument COUNT = true,
skipped here).  Fold passes (pt_fold_kernel) of the timed launches are inside the span.

  python tools/trace_span.py <kernel_trace.csv> <bench_line.json> <out.json> ["note"]
"""
import csv, json, sys

trace, line, out = sys.argv[1], sys.argv[2], sys.argv[3]
note = sys.argv[4] if len(sys.argv) > 4 else ""
d = json.loads([l for l in open(line) if l.startswith("{")][-1])
W, K, F = d["warmup"], d["steps"], d["config"]["frames_in_flight"]
rows = [r for r in csv.DictReader(open(trace))]
path = [r for r in rows if "pt_path_kernel<" in r["Kernel_Name"] and not r["Kernel_Name"].split("pt_path_kernel<")[1].split(",")[1].strip().startswith("true")]
path.sort(key=lambda r: int(r["Start_Timestamp"]))
fold = sorted((r for r in rows if "pt_fold_kernel" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
st = [int(r["Start_Timestamp"]) for r in path]; en = [int(r["End_Timestamp"]) for r in path]
L = int(d["roofline"]["launches"])
n_serial = max(0, min(K - 4, 128)) if F > 1 else 0      # timed frames of each probe
m = 4 + n_serial if n_serial else 0
c = d["config"]
npix_ = c["width"] * c["height"]
ahead_applies = (c["spp_per_step"] < 4 or npix_ > (1 << 22)) and npix_ <= (1 << 24)      # the calls render-ahead takes (launch_serial, pt_api.cpp); else the second probe is m serial launches too
tail = (2 * m + (1 if ahead_applies else 0)) if n_serial else 0
timed = range(len(path) - tail - L, len(path) - tail)
first, last = min(st[i] for i in timed), max(en[i] for i in timed)
folds_in = [r for r in fold if first <= int(r["Start_Timestamp"]) and int(r["Start_Timestamp"]) <= last + 200000]
last_fold = max([int(r["End_Timestamp"]) for r in folds_in] + [last])
busy = sum(en[i] - st[i] for i in timed)
res = {"source": note, "timed_steps": K, "timed_launches": L, "steps_per_launch": K / L, "warmup": W, "frames_in_flight": F,
       "first_start_to_last_end_ns": last - first, "ns_per_launch_amortised": (last - first) / L,
       "first_start_to_last_fold_end_ns": last_fold - first, "ns_per_launch_incl_last_fold": (last_fold - first) / L,
       "ns_per_step_incl_last_fold": (last_fold - first) / K,
       "mean_dispatch_duration_ns": busy / L, "mean_launches_in_flight": busy / (last - first),
       "fold_passes_in_span": len(folds_in), "mean_fold_duration_ns": (sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in folds_in) / max(len(folds_in), 1))}
ser = range(len(path) - tail + 4, len(path) - tail + m)
if len(ser) > 0:
    res["serial_launches"] = len(ser)
    res["serial_ns_per_launch"] = (max(en[i] for i in ser) - min(st[i] for i in ser)) / len(ser)
    res["serial_mean_dispatch_duration_ns"] = sum(en[i] - st[i] for i in ser) / len(ser)
    # the render-ahead probe: frame j of it (j >= 2) was launched ahead during frame j - 1, so the kernels of its timed
    # frames 4 .. m-1 are dispatches 4 .. m-1 of the phase; the timed region ends with the fold pass of the last of them
    ah = range(len(path) - (m + 1) + 4, len(path) - 1) if ahead_applies else range(len(path) - m + 4, len(path))
    a_last = max(en[i] for i in ah)
    res["render_ahead_launches"] = len(ah)
    res["render_ahead_ns_per_frame"] = (a_last - st[ah[0]]) / len(ah)
    res["render_ahead_mean_dispatch_duration_ns"] = sum(en[i] - st[i] for i in ah) / len(ah)
r = d["roofline"]
res["bench_line_of_the_same_run"] = {"value_msamples_per_s": d["value"], "ms_per_step": d["ms_per_step"], "kernel_ms": r["kernel_ms"],
                                     "kernel_ms_serial": r.get("kernel_ms_serial"), "frame_ms_render_ahead": r.get("frame_ms_render_ahead"), "frac": r["frac"], "flops_per_launch": r["flops_per_launch"]}
res["frac_from_this_trace"] = r["flops_per_launch"] / (res["ns_per_launch_incl_last_fold"] * 1e-9) / 1e12 / r["peak"]
res["frac_path_kernels_only"] = r["flops_per_launch"] / (res["ns_per_launch_amortised"] * 1e-9) / 1e12 / r["peak"]
res["frac_line_over_trace"] = r["frac"] / res["frac_from_this_trace"]
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
