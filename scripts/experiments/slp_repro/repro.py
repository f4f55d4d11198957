import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from raytracing_amd import _native as N
N.LIB_PATH = os.environ["TRR_LIB"]
from raytracing_amd import standin
from raytracing_amd.raytracer import Raytracer
from oracle import oracle
sc = standin.parity_scene(320, 240, detail=16, sky_image=False)
osc = oracle.OracleScene(sc)
for md in (1, 3):
    rt = Raytracer(sc, max_depth=md)
    out = rt.render_frame(want_ids=True)
    ref = osc.render(shade=oracle.SHADE_HOISTED, max_depth=md)
    d = np.abs(out['rgb'].astype(int) - ref['rgb'].astype(int)).max(axis=2)
    print(os.path.basename(os.environ["TRR_LIB"]), "max_depth", md, "pixels differing", int((d > 0).sum()), "of", d.size, "max diff", int(d.max()),
          "ids differing", int((out['hit_tri'] != ref['hit_tri']).sum()), flush=True)
    rt.close()
