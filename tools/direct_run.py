"""A few steps of BASELINE config 1 (cpu_default, 1280x720, `direct`, 64 spp per step) for rocprofv3 passes on the FP64
kernel:  rocprofv3 --pmc ... -- python3 tools/direct_run.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from path_tracer_amd import abi, scenes
from path_tracer_amd.renderer import Renderer
r = Renderer(0)
sc, cam = scenes.cpu_default(1280, 720)
r.set_scene(sc); r.set_camera(cam); r.reset()
p = abi.render_params(integrator=abi.INTEGRATOR_DIRECT)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    r.render(i * 64, 64, p)
r.sync()
