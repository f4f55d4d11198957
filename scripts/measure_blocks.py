#!/usr/bin/env python3
"""Cost of the progressive render (trr_render_blocks) against the whole-frame form on the 4K car stand-in: frame time
and time to the first block for several batch sizes.  One JSON line per case.   python scripts/measure_blocks.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import standin  # noqa: E402
from raytracing_amd.raytracer import Raytracer  # noqa: E402


def main():
    for spp in (1, 16):
        sc = standin.standin_car(3840, 2160, n_tris=1_000_000, seed=1, anti_aliasing=spp > 1, samples=spp)
        rt = Raytracer(sc, max_depth=5)
        rt.render_frame()
        t = []
        for _ in range(3):
            t0 = time.perf_counter()
            rt.render_frame()
            t.append((time.perf_counter() - t0) * 1e3)
        print(json.dumps({"spp": spp, "form": "trr_render", "frame_ms": round(min(t), 2)}), flush=True)
        for bpb in (10, 20, 40, 100, 400):
            best = None
            for _ in range(3):
                first = []
                t0 = time.perf_counter()
                rt.render_blocks(lambda xb, yb, blk: first.append(time.perf_counter()) if not first else None, blocks_per_batch=bpb)
                total = (time.perf_counter() - t0) * 1e3
                rec = (total, (first[0] - t0) * 1e3)
                best = rec if best is None or rec[0] < best[0] else best
            print(json.dumps({"spp": spp, "form": "trr_render_blocks", "blocks_per_batch": bpb, "frame_ms": round(best[0], 2),
                              "first_block_ms": round(best[1], 2)}), flush=True)
        rt.close()


if __name__ == "__main__":
    main()
