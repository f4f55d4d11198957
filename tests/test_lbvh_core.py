"""CPU test of the device BVH build's per-element code (raytracing_amd/csrc/lbvh_core.h): the functions the HIP kernels of
build_device.hip call are compiled with g++ and run serially by tests/lbvh_core_check.cpp, which checks the radix
tree (ranges partition, parents consistent), the bottom-up fit (every inner node reached exactly twice) and the
collapsed 4-wide tree (every triangle in exactly one leaf of <= 4, boxes nested), with and without the treelet pass."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("lbvh") / "lbvh_core_check")
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-ffp-contract=off", os.path.join(HERE, "lbvh_core_check.cpp"), "-o", exe],
                   check=True)
    return exe


@pytest.mark.parametrize("n,seed,mode", [(1, 1, 0), (4, 1, 0), (5, 1, 0), (6, 2, 0), (7, 3, 1), (64, 4, 3), (1000, 5, 1), (5000, 6, 2),
                                         (100_000, 7, 0), (100_000, 8, 3), (200_000, 9, 1), (100_000, 1, 4)])
@pytest.mark.parametrize("treelet_max", [0, 8, 128])
def test_serial_run_of_the_kernel_bodies_builds_a_valid_tree(checker, n, seed, mode, treelet_max):
    r = subprocess.run([checker, str(n), str(seed), str(mode), str(treelet_max)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr


def test_rebuilt_treelets_lower_the_surface_area_cost(checker):
    """The point of the treelet pass: on a tessellated surface the inner nodes' summed surface falls, more with bigger treelets."""
    cost = {}
    for t in (0, 128, 2048):
        r = subprocess.run([checker, "100000", "1", "4", str(t)], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr
        cost[t] = float(r.stdout.split("inner_cost=")[1].split()[0])
    assert cost[2048] < cost[128] < 0.97 * cost[0], cost
