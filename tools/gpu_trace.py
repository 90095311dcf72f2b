"""Per-wave timeline of one launch (needs a PT_TRACE kernel build: make variant NAME=trace VFLAGS="-DPT_TRACE").
python tools/gpu_trace.py [few_max|-] [variant library name, default trace] [serial: only the one-frame 720p launch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["PT_AMD_LIB"] = os.path.join(ROOT, "path_tracer_amd", "lib", f"libpt_amd_{sys.argv[2] if len(sys.argv) > 2 else 'trace'}.so")
os.environ["PT_AMD_TRACE"] = "/tmp/pt_trace.bin"
if len(sys.argv) > 1 and sys.argv[1] != "-": os.environ["PT_AMD_FEW_MAX"] = sys.argv[1]
CASES = (("cuda_default", (1280, 720), 1, 1), ("cuda_default", (1280, 720), 1, 20), ("cuda_default", (3840, 2160), 16, 1))
if len(sys.argv) > 3 and sys.argv[3] == "serial": CASES = CASES[:1]
import numpy as np
from path_tracer_amd import abi, scenes
from path_tracer_amd.renderer import Renderer
r = Renderer(0)
for name, (w, h), spp, frames in CASES:
    sc, cam = scenes.builtin(name, w, h)
    r.set_frames_in_flight(1); r.set_scene(sc); r.set_camera(cam); r.reset()
    r.set_frames_in_flight(3 if frames > 1 else 1)      # frames > 1: one launch of that many frames (the last launch is traced)
    for i in range(3 if frames == 1 else frames): r.render(i * spp, spp, abi.render_params(fused_rgba8=1))
    r.sync()
    raw = np.fromfile("/tmp/pt_trace.bin", dtype=np.uint64).reshape(-1, 16 + 128)   # PT_TRACE_WORDS (pt_device.h)
    raw = raw[raw[:, 0] > 0]
    t, ring = raw[:, :16], raw[:, 16:].copy().view(np.uint32)                        # ring: the wave's last 256 trips
    t0 = t[:, 0].min()
    us = lambda x: (x.astype(np.float64) - float(t0)) / 100.0   # 100 MHz
    start, exh, few, end = us(t[:, 0]), us(t[:, 1]), us(t[:, 2]), us(t[:, 3])
    trips, ftrips, frays, ltrips = (t[:, k].astype(np.float64) for k in (4, 5, 6, 7))
    q = lambda x: np.percentile(x, [0, 10, 50, 90, 99, 100]).round(1).tolist()
    print(f"== {name} {w}x{h} {spp} spp x {frames} frame(s): {len(t)} waves, launch span {end.max():.1f} us")
    print("   start            pct[0,10,50,90,99,100]", q(start))
    print("   frame exhausted  ", q(exh[t[:, 1] > 0]))
    print("   first few-trip   ", q(few[t[:, 2] > 0]))
    print("   end              ", q(end))
    print("   end - exhausted  ", q((end - exh)[t[:, 1] > 0]))
    print("   trips/wave       ", q(trips), " few-mode trips/wave", q(ftrips), " few rays/wave", q(frays))
    print(f"   lane utilisation of trips: {ltrips.sum() / (64 * trips.sum()):.3f};  few-mode share of trips {ftrips.sum() / trips.sum():.3f}")
    cyc = t[:, 8:11].astype(np.float64).sum(axis=0)
    print(f"   wave cycles by section: regeneration {cyc[0] / cyc.sum():.3f}, intersection {cyc[1] / cyc.sum():.3f}, shading {cyc[2] / cyc.sum():.3f}"
          f"  ({cyc.sum() / trips.sum():.0f} cycles per wave-trip)")
    fc = t[:, 11:14].astype(np.float64).sum(axis=0)
    if ftrips.sum() > 0:
        print(f"   lane-parallel trips: {fc.sum() / ftrips.sum():.0f} cycles per trip = regeneration {fc[0] / ftrips.sum():.0f} + intersection {fc[1] / ftrips.sum():.0f}"
              f" + shading {fc[2] / ftrips.sum():.0f}; {frays.sum() / ftrips.sum():.2f} rays per trip  (a regeneration stamp of trip k+1 is booked to trip k's mode)")
    busy = end - start
    print(f"   mean trip time normal {((busy.sum()) / trips.sum()):.3f} us")
    late = np.argsort(end)[-5:]
    for i in late:
        fe = (end[i] - few[i]) if t[i, 2] > 0 else 0.0
        print(f"     late wave: end {end[i]:.1f} exh {exh[i]:.1f} few-start {few[i]:.1f} trips {int(trips[i])} few-trips {int(ftrips[i])} -> {fe / max(ftrips[i], 1):.3f} us per few-trip")
        ft = max(ftrips[i], 1); nt = max(trips[i] - ftrips[i], 1)
        tc, tf = t[i, 8:11].astype(np.float64), t[i, 11:14].astype(np.float64)
        print(f"        cycles per few-trip: regen {tf[0] / ft:.0f} isect {tf[1] / ft:.0f} shade {tf[2] / ft:.0f};  per normal trip: regen {(tc[0] - tf[0]) / nt:.0f} isect {(tc[1] - tf[1]) / nt:.0f} shade {(tc[2] - tf[2]) / nt:.0f};"
              f"  all cycles {tc.sum():.0f} over {end[i] - start[i]:.1f} us = {tc.sum() / max(end[i] - start[i], 1e-9) / 1e3:.2f} GHz; few rays {int(frays[i])}")
    # the last waves trip by trip: when each trip began (us since the launch began) and how many lanes traced in it
    for i in late[-3:]:
        n = int(trips[i]); k = min(n, 256)
        e = np.array([ring[i][j & 255] for j in range(n - k, n)], dtype=np.uint64)
        when, lanes = start[i] + (e >> np.uint64(7)).astype(np.float64) / 100.0, (e & np.uint64(127)).astype(int)
        dur = np.diff(np.append(when, end[i]))
        print(f"     trips of the wave that ends at {end[i]:.1f} us (start us : lanes : duration us), last {min(k, 72)} of {n}:")
        print("       " + "  ".join(f"{w:.0f}:{l}:{d:.1f}" for w, l, d in list(zip(when, lanes, dur))[-72:]))
        for lo, hi in ((33, 64), (9, 32), (3, 8), (1, 2)):
            m = (lanes >= lo) & (lanes <= hi)
            if m.any(): print(f"       trips with {lo}-{hi} lanes: {int(m.sum())}, mean {dur[m].mean():.2f} us, median {np.median(dur[m]):.2f}")
    # over all waves: mean trip duration by lanes alive and by when the trip began (tenths of the span)
    allw, alll, alld = [], [], []
    for i in range(len(t)):
        n = int(trips[i]); k = min(n, 256)
        if k == 0: continue
        e = np.array([ring[i][j & 255] for j in range(n - k, n)], dtype=np.uint64) if n > 256 else ring[i][:k].astype(np.uint64)
        w = start[i] + (e >> np.uint64(7)).astype(np.float64) / 100.0
        allw.append(w); alll.append((e & np.uint64(127)).astype(int)); alld.append(np.diff(np.append(w, end[i])))
        if len(allw) >= 3000: break
    allw, alll, alld = np.concatenate(allw), np.concatenate(alll), np.concatenate(alld)
    edges = np.linspace(0, end.max(), 11)
    print("   mean trip duration (us) / mean lanes alive, by tenth of the span in which the trip began (first 3000 waves):")
    print("      " + "  ".join(f"{alld[(allw >= a_) & (allw < b_)].mean():.2f}/{alll[(allw >= a_) & (allw < b_)].mean():.0f}" if ((allw >= a_) & (allw < b_)).any() else "-" for a_, b_ in zip(edges[:-1], edges[1:])))
    # how much of the chip is busy over time
    grid = np.linspace(0, end.max(), 21)
    act = [(int(((start <= g) & (end > g)).sum())) for g in grid]
    print("   active waves over time:", act)
