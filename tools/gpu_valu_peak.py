"""FP32 VALU ceiling micro-benchmark (pt_valu_peak): plain v_fma_f32 vs v_pk_fma_f32, several runs."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from path_tracer_amd.renderer import Renderer
r = Renderer(0)
out = {"kernel": "pt_valu_peak_kernel: 16 independent FMA chains per lane, 8 blocks of 256 threads per CU", "spec_peak_tflops": 157.3,
       "v_fma_f32": [round(r.valu_peak(0, 200000), 2) for _ in range(5)], "v_pk_fma_f32": [round(r.valu_peak(1, 200000), 2) for _ in range(5)]}
out["implied_clock_ghz_from_v_fma"] = round(max(out["v_fma_f32"]) * 1e12 / (256 * 4 * 32 * 2) / 1e9, 3)   # 256 CUs x 4 SIMD-32 x 2 flop
print(json.dumps(out))
