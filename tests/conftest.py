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
