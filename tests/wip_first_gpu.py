import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from raytracing_amd import standin
from raytracing_amd.raytracer import Raytracer
from oracle import oracle
from tests._util import compare_images, compare_ids
for sky in (True, False):
  sc = standin.parity_scene(320, 240, detail=16, sky_image=sky)
  osc = oracle.OracleScene(sc)
  for md in (1,2,3,5):
    rt = Raytracer(sc, max_depth=md)
    out = rt.render_frame(want_ids=True)
    ref = osc.render(shade=oracle.SHADE_HOISTED, max_depth=md)
    print("sky", sky, "max_depth", md, "img within1, exact, max:", compare_images(out['rgb'], ref['rgb']), flush=True)
    rt.close()
