"""gpurun_out/pmc_4k_{a,b,c}.json (tools/pmc_4k.sh) -> profiles/<prefix>_pmc_4k.json with the derived per-trip figures.
python tools/condense_pmc_4k.py r2 "<note on the kernel's commit>" """
import json, sys
prefix, note = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
a, b, c = (json.load(open(f"gpurun_out/pmc_4k_{p}.json")) for p in "abc")
rays = 3840 * 2160 * 32 * 6.4609   # rays per sample of cuda_default at depth 50 (event counters, bench line)
trips = rays / 64.0
valu = a["SQ_INSTS_VALU"]
out = {"source": "tools/pmc_4k.sh: rocprofv3 --pmc, three passes, tools/pt_headless --width 3840 --height 2160 --spp 32 --spp-per-launch 16 "
                 "(cuda_default, 2 launches of pt_path_kernel<false,false,false>); sums over both launches; " + note,
       "pass_a": a, "pass_b": b, "pass_c": c,
       "derived": {"valu_per_full_64_ray_trip": valu / trips, "salu_per_trip": a["SQ_INSTS_SALU"] / trips, "branches_per_trip": b["SQ_INSTS_BRANCH"] / trips,
                   "valu_thread_utilisation": a["SQ_THREAD_CYCLES_VALU"] / (64.0 * valu), "rays": rays,
                   "waves_issuing_frac": b["SQ_ACTIVE_INST_ANY"] / a["SQ_WAVE_CYCLES"],
                   "waves_waitcnt_frac": b["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"],
                   "valu_pipe_busy_at_2_cycles_per_instruction": a["SQ_ACTIVE_INST_VALU"] / 4.0 * 2.0 / (a["SQ_BUSY_CYCLES"] / 8.0 * 4.0) if False else None,
                   "mix": {k[len("SQ_INSTS_VALU_"):]: round(v / valu, 4) for k, v in c.items() if k.startswith("SQ_INSTS_VALU_")},
                   "earlier": {"round_1": {"valu_per_full_64_ray_trip": 697.6, "valu_thread_utilisation": 0.591, "salu_per_trip": 252.8},
                               "round_2_after_regeneration_change": {"valu_per_full_64_ray_trip": 641.2, "valu_thread_utilisation": 0.640, "salu_per_trip": 278.2}}}}
del out["derived"]["valu_pipe_busy_at_2_cycles_per_instruction"]
json.dump(out, open(f"profiles/{prefix}_pmc_4k.json", "w"), indent=1)
print(json.dumps(out["derived"], indent=1))
