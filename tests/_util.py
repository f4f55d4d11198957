"""Shared helpers of the parity tests."""
import numpy as np


def compare_images(gpu_rgb, ref_rgb):
    """Returns (fraction of pixels whose 3 channels are all within 1 LSB, fraction bit-identical, max abs diff)."""
    d = np.abs(gpu_rgb.astype(np.int16) - ref_rgb.astype(np.int16)).max(axis=-1)
    return float((d <= 1).mean()), float((d == 0).mean()), int(d.max())


def compare_ids(g_obj, g_tri, r_obj, r_tri):
    same = (g_obj == r_obj) & (g_tri == r_tri)
    return float(same.mean())
