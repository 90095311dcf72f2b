"""Replays one seed of tests/test_frames_gpu.py::test_render_ahead_under_random_host_behaviour, comparing after EVERY step and
printing the steps: python tools/ahead_replay.py <seed>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from path_tracer_amd import abi, scenes
from path_tracer_amd.renderer import Renderer
torch.cuda.set_stream(torch.cuda.Stream())
a, b = Renderer(0), Renderer(0)
b.set_render_ahead(False)
every = len(sys.argv) > 3 and sys.argv[3] == "every"
for seed in range(int(sys.argv[1]), int(sys.argv[2]) + 1):
  rng = np.random.default_rng(20261111 + seed)
  size = (96, 54); scene_name = "cuda_default"; bound = [None, None]; log = []
  for r in (a, b):
    if r._bound is not None: r.unbind()
    sc, cam = scenes.builtin(scene_name, *size); r.set_scene(sc); r.set_camera(cam); r.reset()
  prm = abi.render_params(fused_rgba8=1); s, n = 0, 1
  for step in range(260):
      k = int(rng.integers(0, 100))
      if k < 62:
          for r in (a, b): r.render(s, n, prm)
          log.append(f"{step}: render {s} x{n}"); s += n if prm.acc_mode == abi.ACC_SUM else 1
      elif k < 66: s = int(rng.integers(0, 50)); log.append(f"{step}: jump to {s}")
      elif k < 70:
          mean = rng.integers(0, 3) == 0; n = 1 if mean else int(rng.integers(1, 4))
          prm = abi.render_params(fused_rgba8=int(rng.integers(0, 2)), acc_mode=abi.ACC_RUNNING_MEAN if mean else abi.ACC_SUM,
                                  flags=abi.FLAG_NEE_MIS if rng.integers(0, 3) == 0 else 0, max_depth=int(rng.choice([2, 8, 50])))
          if mean:
              for r in (a, b): r.reset()
              s = 0
          log.append(f"{step}: params mean={mean} n={n} fused={prm.fused_rgba8} flags={prm.flags} depth={prm.max_depth}")
      elif k < 74:
          for r in (a, b): r.reset()
          log.append(f"{step}: reset")
      elif k < 78:
          if rng.integers(0, 2): size = [(96, 54), (120, 40), (64, 64)][int(rng.integers(0, 3))]
          eye = (float(rng.uniform(-0.5, 0.5)), 1.0 + float(rng.uniform(-0.3, 0.3)), 3.0)
          cam = abi.make_camera(eye, (0, 1, 0), (0, 1, 0), 45.0, size[0], size[1], abi.CAMERA_CUDA)
          for i, r in enumerate((a, b)):
              if bound[i] is not None: r.unbind(); bound[i] = None
              r.set_camera(cam); r.reset()
          log.append(f"{step}: camera {size}")
      elif k < 80:
          scene_name = ["cuda_default", "glass_stress"][int(rng.integers(0, 2))]
          for r in (a, b):
              sc, _ = scenes.builtin(scene_name, *size); r.set_scene(sc); r.reset()
          log.append(f"{step}: scene {scene_name}")
      elif k < 83:
          for i, r in enumerate((a, b)):
              if bound[i] is None:
                  t = torch.zeros((size[1], size[0], 4), device="cuda", dtype=torch.float32); torch.cuda.synchronize()
                  r.bind_tensor(t); bound[i] = t
              else:
                  r.unbind(); bound[i] = None; r.reset()
          log.append(f"{step}: bind/unbind -> bound={bound[0] is not None}")
      elif k < 86:
          which = []
          for r in (a, b):
              t_ = bool(rng.integers(0, 2)); which.append(t_); (r.use_torch_stream if t_ else r.use_own_stream)()
          log.append(f"{step}: streams torch={which}")
      elif k < 89:
          for r in (a, b): r.set_frames_in_flight(2)
          m = int(rng.integers(1, 6))
          for _ in range(m):
              for r in (a, b): r.render(s, n, prm)
              s += n if prm.acc_mode == abi.ACC_SUM else 1
          for r in (a, b): r.set_frames_in_flight(1)
          log.append(f"{step}: {m} frames in flight")
      elif k < 92:
          for r in (a, b): r.upload_accum(r.accum())
          log.append(f"{step}: checkpoint")
      else: log.append(f"{step}: look")
      if not np.array_equal(a.accum(), b.accum()):
          print("\n".join(log[-25:])); print("FIRST DIFFERENCE after step", step, "w:", np.unique(a.accum()[..., 3]), np.unique(b.accum()[..., 3])); break
  else:
    print("seed", seed, "no difference", flush=True)
