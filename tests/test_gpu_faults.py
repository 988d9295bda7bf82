"""GPU: what the kernels' waits do when the thing they wait for never comes, exercised ONCE each through a test-only build of the same sources
(jittor-myc-nerfs_amd/lib/variants/libtvr_faults.so, built by csrc/Makefile with -DTVR_FAULT_INJECT_MARCH -DTVR_DEBUG_SIMD, selected per
process with TVR_LIB_PATH — a library is loaded once per process, so each case runs in a child).

  * march: the dynamic tile queue's waiters are bounded; a waiter that was overtaken raises the fault flag in the scratch header, the grid
    drains, and the composite kernel writes NaN to every pixel and depth of the call (tvr_march.hip, include/tvr.h tvr_scratch_layout).
  * shade: the matrix token pairs waves w and w + 4 of a workgroup on the assumption that they share a SIMD (tvr_shade.hip); the debug build
    reads HW_ID and counts mismatching pairs.
"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

FAULTS_LIB = os.path.join(ROOT, "jittor-myc-nerfs_amd", "lib", "variants", "libtvr_faults.so")

CHILD = r"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
from conftest import GOLDEN, TINY, make_model
from jittor_myc_nerfs_amd import _lib as L, synthetic
import ctypes as C
dump = dict(np.load(os.path.join(GOLDEN, "tiny_dump.npz")))
arrs = {{k[len("scene."):]: v for k, v in dump.items() if k.startswith("scene.")}}
hyper = dict(synthetic.HYPER, near_far=TINY["near_far"], step_ratio=TINY["step_ratio"])
m = make_model(arrs, hyper)
rays = torch.tensor(np.concatenate([dump["rays"]] * 1024), device="cuda")          # 65 536 rays: 4096 tiles of 16, ~16 per workgroup
stats = torch.zeros(24, dtype=torch.int64, device="cuda")
rgb, depth = m.render_rays(rays, white_bg=True, N_samples=TINY["N_samples"], stats=stats)
torch.cuda.synchronize()
lay = L.ScratchLayout()
L.check(L.lib().tvr_scratch_describe(rays.shape[0], TINY["N_samples"], C.byref(lay)), "describe")
hdr = m._scratch[lay.counter:lay.counter + 16].view(torch.int32).tolist()
st = stats.tolist()
print("RESULT " + json.dumps(dict(fault=hdr[2], nan_rgb=int(torch.isnan(rgb).sum()), nan_depth=int(torch.isnan(depth).sum()), n=int(rays.shape[0]),
                                  simd_mismatch=st[15], simd_ids=st[16:24])))
"""


# the TRAINING entry point: tvr_march_forward puts the queue into ray order behind the march (tvr_step.hip: scan, gather, copy back).  A faulted march
# leaves ray_off / ray_cnt of some rays unwritten; here the scratch is poisoned with 0xFF first, so those words read 4 294 967 295: the three queue kernels must return
# at once on the fault flag (and clamp besides), or they read and write gigabytes out of bounds.
CHILD_TRAIN = r"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
from conftest import GOLDEN, TINY, make_model
from jittor_myc_nerfs_amd import _lib as L, synthetic
from jittor_myc_nerfs_amd.autograd_ops import _stream_ptr
import ctypes as C
dump = dict(np.load(os.path.join(GOLDEN, "tiny_dump.npz")))
arrs = {{k[len("scene."):]: v for k, v in dump.items() if k.startswith("scene.")}}
hyper = dict(synthetic.HYPER, near_far=TINY["near_far"], step_ratio=TINY["step_ratio"])
m = make_model(arrs, hyper)
rays = torch.tensor(np.concatenate([dump["rays"]] * 1024), device="cuda")          # 65 536 rays: the largest batch the ray-order pass takes
n, S = int(rays.shape[0]), TINY["N_samples"]
sc = m._ensure_scene(force=True)
lay = L.ScratchLayout()
L.check(L.lib().tvr_scratch_describe(n, S, C.byref(lay)), "describe")
scratch = torch.full((lay.total,), 0xFF, dtype=torch.uint8, device="cuda")
depth = torch.empty(n, dtype=torch.float32, device="cuda")
rc = L.lib().tvr_march_forward(sc, rays.data_ptr(), n, S, None, 1e-4, depth.data_ptr(), scratch.data_ptr(), scratch.numel(), _stream_ptr(m.device))
torch.cuda.synchronize()
hdr = scratch[lay.counter:lay.counter + 16].view(torch.int32).tolist()
# a second call on the same (now partly written) scratch, and an allocation + sync behind it: a stray write would have landed somewhere by now
rc2 = L.lib().tvr_march_forward(sc, rays.data_ptr(), n, S, None, 1e-4, depth.data_ptr(), scratch.data_ptr(), scratch.numel(), _stream_ptr(m.device))
probe = torch.zeros(1 << 20, device="cuda"); torch.cuda.synchronize()
print("RESULT " + json.dumps(dict(rc=rc, rc2=rc2, fault=hdr[2], qlen=hdr[0] & 0xffffffff, cap=n * S, probe=float(probe.sum()))))
"""


def _run_child(lib_path, child=CHILD):
    env = dict(os.environ, TVR_LIB_PATH=lib_path)
    r = subprocess.run([sys.executable, "-c", child.format(root=ROOT)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_march_tile_queue_miss_is_loud_and_the_grid_drains():
    assert os.path.exists(FAULTS_LIB), "csrc/Makefile builds lib/variants/libtvr_faults.so (make -C jittor-myc-nerfs_amd/csrc)"
    d = _run_child(FAULTS_LIB)
    # the process came back (no hang), the flag says "overtaken" (1), and no pixel of the call pretends to be a result
    assert d["fault"] == 1
    assert d["nan_rgb"] == 3 * d["n"] and d["nan_depth"] == d["n"]
    # the same build's shade kernel: waves w and w + 4 of every workgroup report the same SIMD id
    assert d["simd_mismatch"] == 0, d
    assert d["simd_ids"][:4] == d["simd_ids"][4:], d


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_product_library_raises_no_fault_flag():
    from jittor_myc_nerfs_amd import _lib as L
    d = _run_child(L.LIB_PATH)
    assert d["fault"] == 0 and d["nan_rgb"] == 0 and d["nan_depth"] == 0


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_training_march_fault_leaves_the_ray_order_pass_inert():
    """tvr_march_forward on poisoned scratch with the injected tile-queue miss: the flag is raised, the queue pass behind the march touches nothing, the process lives."""
    assert os.path.exists(FAULTS_LIB)
    d = _run_child(FAULTS_LIB, CHILD_TRAIN)
    assert d["rc"] == 0 and d["rc2"] == 0 and d["fault"] == 1 and d["probe"] == 0.0
    assert d["qlen"] <= d["cap"]
