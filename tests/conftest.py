import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _guard_bytes_behind_every_buffer():
    """TVR_GUARDS=1 (an opt-in sweep: `TVR_GUARDS=1 pytest tests -m gpu`): EVERY test of the suite runs with 4 KB of 0xA5 behind each caller-owned buffer the Python host
    hands to the library (_lib.dev_bytes / dev_empty), and the guards are checked after each test — tests/test_gpu_canaries.py's mechanism over the whole suite (round 6: it
    adds the fused step's wider workspaces, the column-block X, the cropped gradients of narrow networks, the pieces' two scratch halves under every test that renders)."""
    if os.environ.get("TVR_GUARDS", "0") in ("", "0"):
        yield
        return
    from jittor_myc_nerfs_amd import _lib as L
    keep = L.GUARD_BYTES
    L._guarded.clear()
    L.GUARD_BYTES = 4096
    yield
    import torch
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    bad = L.check_guards()
    L.GUARD_BYTES = keep
    L._guarded.clear()
    assert bad == [], f"a kernel wrote behind a caller-owned buffer: {bad}"


def _has_gpu():
    import torch
    return torch.cuda.is_available()


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container (run -m gpu through gpurun)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


TINY = dict(gridSize=[16, 20, 24], aabb=[[-1.5, -1.2, -1.0], [1.5, 1.2, 1.0]], near_far=[2.0, 6.0], step_ratio=0.5,
            N_samples=48)


@pytest.fixture(scope="session")
def hyper_tiny():
    from jittor_myc_nerfs_amd import synthetic
    return dict(synthetic.HYPER, near_far=TINY["near_far"], step_ratio=TINY["step_ratio"])


@pytest.fixture(scope="session")
def tiny_dump():
    return dict(np.load(os.path.join(GOLDEN, "tiny_dump.npz")))


@pytest.fixture(scope="session")
def tiny_edge():
    return dict(np.load(os.path.join(GOLDEN, "tiny_edge.npz")))


@pytest.fixture(scope="session")
def config1_golden():
    return dict(np.load(os.path.join(GOLDEN, "config1.npz")))


@pytest.fixture(scope="session")
def tiny_arrays(tiny_dump):
    return {k[len("scene."):]: v for k, v in tiny_dump.items() if k.startswith("scene.")}


@pytest.fixture(scope="session")
def tiny_ref():
    return dict(np.load(os.path.join(GOLDEN, "tiny_ref.npz")))


@pytest.fixture(scope="session")
def tiny_ref_arrays(tiny_arrays, tiny_ref):
    """REFTensoRF parameters: the tiny scene's VM factors / basis / W2 / W3 plus the arrays tiny_ref.npz adds or replaces."""
    a = dict(tiny_arrays)
    a.update({k[len("scene."):]: v for k, v in tiny_ref.items() if k.startswith("scene.")})
    return a


@pytest.fixture(scope="session")
def tiny_npp():
    return dict(np.load(os.path.join(GOLDEN, "tiny_npp.npz")))


@pytest.fixture(scope="session")
def tiny_npp_arrays(tiny_arrays, tiny_npp):
    """NerfPlusPlus parameters: the tiny TensorVMSplit scene plus the background network of tiny_npp.npz."""
    a = dict(tiny_arrays)
    a.update({k[len("scene."):]: v for k, v in tiny_npp.items() if k.startswith("scene.")})
    return a


def make_model(arrs, hyper, device="cuda", gridSize=None, aabb=None):
    """TensorVMSplit (REFTensoRF when the arrays hold its extra linears) on `device` holding the given arrays (reference
    constructor signature, train.py:167-172)."""
    from jittor_myc_nerfs_amd import NerfPlusPlus, REFTensoRF, TensorVMSplit
    cls = REFTensoRF if "normal_W" in arrs else (NerfPlusPlus if "bg.radii" in arrs else TensorVMSplit)
    m = cls(arrs["aabb"] if aabb is None else aabb, [int(x) for x in (arrs["gridSize"] if gridSize is None else gridSize)],
                      device, density_n_comp=[16, 16, 16], appearance_n_comp=[48, 48, 48], app_dim=27,
                      near_far=hyper["near_far"], shadingMode="MLP_Fea", alphaMask_thres=1e-4,
                      density_shift=hyper["density_shift"], distance_scale=hyper["distance_scale"],
                      rayMarch_weight_thres=hyper["rayMarch_weight_thres"], pos_pe=6, view_pe=2, fea_pe=2, featureC=128,
                      step_ratio=hyper["step_ratio"], fea2denseAct=hyper["fea2denseAct"])
    m.load_arrays(arrs)
    return m


# ---------------------------------------------------------------------------------------------------------------------
# alt path: seeded Instant-NGP scene shared by the oracle tests and the GPU parity tests
NGP_AABB_SCALE = 4


@pytest.fixture(scope="session")
def ngp_scene():
    """(levels, arrays): synthetic.make_ngp_scene_arrays at aabb_scale 4 plus the oracle's bitfield of its density grid."""
    from jittor_myc_nerfs_amd import synthetic
    from oracle import ngp_oracle as N
    levels = N.grid_levels(NGP_AABB_SCALE)
    arrs = synthetic.make_ngp_scene_arrays(levels["offsets"])
    arrs["density_grid_bitfield"], arrs["density_grid_mean"] = N.update_bitfield(arrs["density_grid"])
    return levels, arrs


def ngp_camera_rays(W, H, pose_index=0, radius=4.0):
    """Blender-convention pose on a sphere -> NGP rays through the oracle's restatement of dataset.py."""
    import math
    from jittor_myc_nerfs_amd import rays as R
    from oracle import ngp_oracle as N
    pose = R.sphere_poses(8, radius)[pose_index]
    focal = 0.5 * W / math.tan(0.5 * 0.6911)
    return N.generate_rays(N.matrix_nerf2ngp(pose), W, H, (focal, focal))


def ngp_edge_rays():
    """Rays the reference's sampler treats specially: a zero direction component (division by zero in the slab test), an origin
    inside the occupied shell, a ray that misses the box, the render loop's padding ray (o = d = 1), and an axis-aligned ray."""
    o = np.array([[0.5, 0.5, -1.2], [0.5, 0.7, 0.5], [3.5, 3.5, 3.5], [1.0, 1.0, 1.0], [-1.4, 0.52, 0.49], [0.5, 0.5, 2.4]], np.float32)
    d = np.array([[0.0, 0.0, 1.0], [0.6, 0.0, 0.8], [1.0, 0.0, 0.0], [1.0, 1.0, 1.0], [1.0, 0.0, 0.0], [0.0, 0.6, -0.8]], np.float32)
    return o, d
