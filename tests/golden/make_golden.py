#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the CPU oracle (oracle/trayracer_oracle.cpp).

The reference (Rust, un-vendored crates) cannot be built or run here and ships no fixtures of its own
(SURVEY.md section 4), so these vectors are OUR oracle's outputs: they pin the oracle and the HIP path against
silent drift, they do not pin either against the reference ("parity unpinned").  Each fixture
carries its full input scene (geometry, materials, textures, lights, camera) so it does not depend on the
generator's numpy transcendental loops being bit-reproducible across CPUs.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import oracle  # noqa: E402
from raytracing_amd import standin  # noqa: E402
from tests._util import scene_to_arrays  # noqa: E402


def main():
    # 1. the parity scene (every branch of the path), literal per-light recursion, max_depth 3
    sc = standin.parity_scene(80, 60, detail=8)
    r = oracle.OracleScene(sc).render(shade=oracle.SHADE_LITERAL, max_depth=3, threads=1, want_f32=True)
    np.savez_compressed(os.path.join(HERE, "parity_80x60_d8_md3.npz"), rgb=r["rgb"], hit_obj=r["hit_obj"],
                        hit_tri=r["hit_tri"], rgb_f32=r["rgb_f32"], **scene_to_arrays(sc))
    # 2. the same scene with anti-aliasing (counter RNG, 4 samples, seed 7)
    sc = standin.parity_scene(60, 40, detail=6)
    sc.settings.anti_aliasing, sc.settings.samples, sc.settings.seed = True, 4, 7
    r = oracle.OracleScene(sc).render(shade=oracle.SHADE_LITERAL, max_depth=2, threads=1, want_ids=False)
    np.savez_compressed(os.path.join(HERE, "parity_aa4_60x40_d6_md2.npz"), rgb=r["rgb"], **scene_to_arrays(sc))
    # 3. counter RNG known answers
    keys = [(1, 0, 0, 0), (1, 0, 0, 1), (1, 12345, 3, 0), (7, 4_000_000, 15, 1), (2**63 + 5, 99, 2, 0)]
    vals = np.array([oracle.rand_f32(*k) for k in keys], np.float32)
    np.savez_compressed(os.path.join(HERE, "rand_f32.npz"), keys=np.array(keys, np.uint64), values=vals)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
