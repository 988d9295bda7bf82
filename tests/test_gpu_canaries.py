"""GPU (-m gpu): guard-byte canaries behind every caller-owned buffer.

include/tvr.h: "ALL device memory is caller-owned ... sizes come from the *_bytes() queries".  A kernel that writes past the size it was told — round 2's
`h [m, 96]` where the kernel writes 144 columns, round 5's ticket word of bg_mlp_kernel, which a working-tree build placed on the block table that
tvr_mlpnet_repack walks (gpurun_out/r5_npp_crash.txt: `Fatal Python error: Aborted` in the next training step's backward) — corrupts whatever the
allocator put behind it and shows up somewhere else, one step later, or not at all.  Here every such buffer (`_lib.dev_bytes` / `dev_empty`: scratch, work,
packed images, gradient scratch, outputs with a `*_bytes` argument) gets 4 KB of 0xA5 behind its last byte, the render / training / background paths run, and
the guards must be intact; images the header calls read-only must come back bit for bit."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import TINY, make_model

pytestmark = pytest.mark.gpu


@pytest.fixture
def guards():
    from jittor_myc_nerfs_amd import _lib as L
    L._guarded.clear()
    L.GUARD_BYTES = 4096
    yield L
    L.GUARD_BYTES = 0
    L._guarded.clear()


def _names(L):
    return sorted({w for _, _, w in L._guarded})


def test_the_canary_mechanism_sees_an_overrun(guards):
    L = guards
    t = L.dev_bytes(1000, "cuda", what="victim")
    assert t.numel() == 1000 and t.data_ptr() % 256 == 0 and L.check_guards() == []
    base = L._guarded[-1][0]
    base[1003] = 7                                    # what a kernel writing 4 bytes too far does
    assert L.check_guards() == [("victim", 1000, 3)]
    f = L.dev_empty((5, 3), torch.float32, "cuda", "f32 victim")
    assert f.shape == (5, 3) and f.dtype == torch.float32 and f.is_contiguous()


def test_render_and_training_write_nothing_behind_their_buffers(guards, tiny_dump, tiny_arrays, tiny_ref_arrays, hyper_tiny):
    """tvr_render (scratch sized by tvr_render_scratch_bytes, the header words the kernels own included), a second frame's scratch slot, the dense outputs' path,
    the fused training step (tvr_train_forward / _backward: scratch, work, gradient scratch, the training image), the eager chain (tvr_march_forward's own
    scratch, tvr_gemm_tn / tvr_colsum scratch and outputs), the three fused regularisers — TensorVMSplit and REFTensoRF."""
    from jittor_myc_nerfs_amd import losses
    L = guards
    rays = torch.tensor(tiny_dump["rays"], device="cuda")
    for arrs in (tiny_arrays, tiny_ref_arrays):
        m = make_model(arrs, hyper_tiny)
        a, _ = m.render_rays(rays, white_bg=True, N_samples=TINY["N_samples"])
        b, _, _ = m.render_rays(rays, white_bg=True, N_samples=TINY["N_samples"], dense=True, scratch_slot=1)
        assert torch.equal(a, b)
        m.mlp_arith = "f16"                           # the gate's probe renders and the fp16 copies inside the packed scene
        m.render_rays(rays, white_bg=True, N_samples=TINY["N_samples"])
        m.mlp_arith = "f32"
        for static in (True, False):
            m.static_training = static
            for p in m.parameters():
                p.grad = None
            rgb, _ = m.render_rays_autograd(rays, white_bg=True, N_samples=TINY["N_samples"])
            loss = (rgb ** 2).mean() + 1e-3 * m.density_L1() + 1e-3 * m.vector_comp_diffs()
            reg = losses.TVLoss()
            loss = loss + 1e-2 * m.TV_loss_density(reg) + 1e-2 * m.TV_loss_app(reg)
            loss.backward()
        torch.cuda.synchronize()
        assert L.check_guards() == [], L.check_guards()
    # round 6: a scene with TensorBase's own six encoding frequencies — the fused step's wider workspace (X as three column blocks, the [128,152] temporary, the
    # streamed W1^T image behind the LDS image of the training image buffer) and the lockstep render path
    from jittor_myc_nerfs_amd import TensorVMSplit, synthetic
    arrs6 = synthetic.make_scene_arrays(TINY["gridSize"], TINY["aabb"], seed=5, view_pe=6, fea_pe=6)
    m = TensorVMSplit(arrs6["aabb"], [int(x) for x in arrs6["gridSize"]], "cuda", density_n_comp=[16, 16, 16], appearance_n_comp=[48, 48, 48], app_dim=27,
                      near_far=hyper_tiny["near_far"], shadingMode="MLP_Fea", alphaMask_thres=1e-4, density_shift=hyper_tiny["density_shift"],
                      distance_scale=hyper_tiny["distance_scale"], rayMarch_weight_thres=hyper_tiny["rayMarch_weight_thres"], pos_pe=6, view_pe=6, fea_pe=6,
                      featureC=128, step_ratio=hyper_tiny["step_ratio"], fea2denseAct=hyper_tiny["fea2denseAct"])
    m.load_arrays(arrs6)
    m.render_rays(rays, white_bg=True, N_samples=TINY["N_samples"])
    for static in (True, False):
        m.static_training = static
        rgb, _ = m.render_rays_autograd(rays, white_bg=True, N_samples=TINY["N_samples"])
        assert type(rgb.grad_fn).__name__.startswith("_FusedStepFn") == static
        (rgb ** 2).mean().backward()
    torch.cuda.synchronize()
    assert L.check_guards() == [], L.check_guards()
    names = _names(L)
    print(f"{len(L._guarded)} guarded buffers: {names}")
    for want in ("tvr_scene packed", "tvr_render scratch", "tvr_render scratch (slot 1)", "tvr_render rgb_out", "tvr_train work", "tvr_train_forward scratch",
                 "tvr_grad_scratch", "tvr_march_forward scratch", "tvr_tv_loss scratch", "tvr_l1_mean scratch", "tvr_line_ortho scratch"):
        assert want in names, want


def test_background_network_buffers_and_the_const_image(guards, tiny_npp, tiny_npp_arrays, hyper_tiny):
    """NerfPlusPlus's background path — the one whose ticket word caused round 5's abort: inference forward (the packed image must come back BIT FOR BIT: `const`
    means const since TVR_VERSION 140, the ticket word lives in the caller's work buffer), then the training loop that aborted (pack once, repack + train forward +
    backward + fused Adam per step: the block table inside the training image must survive every forward, or the next repack reads through wild pointers), then
    the same parameters packed again: nothing but the fragment / bias bytes the optimizer moved may differ."""
    from jittor_myc_nerfs_amd import OctreeRender_trilinear_fast
    L = guards
    rays = torch.tensor(tiny_npp["rays"], device="cuda")
    rf, rb = torch.tensor(tiny_npp["rand_fg"], device="cuda"), torch.tensor(tiny_npp["rand_bg"], device="cuda")
    m = make_model(tiny_npp_arrays, hyper_tiny)
    with torch.no_grad():
        a, _ = m(rays, is_train=False, N_samples=TINY["N_samples"], rand_fg=rf, rand_bg=rb)
    img0 = m._bg_image.clone()
    with torch.no_grad():
        for mode in ("f32", "f16act", "f16"):         # three instantiations of bg_mlp_kernel, the gate's probe forwards
            m.mlp_arith = mode
            m(rays, is_train=False, N_samples=TINY["N_samples"], rand_fg=rf, rand_bg=rb)
        m.mlp_arith = "f32"
        b, _ = m(rays, is_train=False, N_samples=TINY["N_samples"], rand_fg=rf, rand_bg=rb)
    torch.cuda.synchronize()
    assert torch.equal(m._bg_image, img0), "a forward wrote into the packed network it takes as const"
    assert torch.equal(a, b)
    assert int(m._bg_work().view(torch.int32)[0]) > 0           # the ticket word is where the header says: in the work buffer
    gt = torch.tensor(tiny_npp["out.rgb_map"], device="cuda").roll(1, dims=1)
    opt = torch.optim.Adam(m.get_optparam_groups(0.02, 0.001), betas=(0.9, 0.99))
    table = None
    for it in range(6):
        opt.zero_grad()
        rgb_map, _, _, _, _ = OctreeRender_trilinear_fast(rays, m, chunk=4096, N_samples=TINY["N_samples"], white_bg=False, is_train=True)
        torch.mean((rgb_map - gt) ** 2).backward()
        opt.step()
        st = m._bg_tstate
        # the block table (include/tvr.h tvr_mlpnet_layout: written by tvr_mlpnet_pack once, walked by tvr_mlpnet_repack on the device): bit-identical after every step
        lay = L.MlpnetLayout()
        L.check(L.lib().tvr_mlpnet_describe(C.byref(m._bg_kernel_desc()), C.byref(lay)), "tvr_mlpnet_describe")
        assert lay.total == st["image"].numel() and lay.block_table_bytes % 40 == 0
        tab = st["image"][lay.block_table: lay.block_table + lay.block_table_bytes].clone()
        if table is None:
            table = tab
            ptrs = tab.view(torch.int64).view(-1, 5)[:, 0]
            assert bool((ptrs != 0).all()), "block table not where this test computes it"
        assert torch.equal(tab, table), f"training step {it}: the block table inside the packed training image changed"
    torch.cuda.synchronize()
    assert L.check_guards() == [], L.check_guards()
    names = _names(L)
    print(f"{len(L._guarded)} guarded buffers: {names}")
    for want in ("tvr_mlpnet packed (inference)", "tvr_mlpnet packed (training)", "tvr_mlpnet work", "tvr_mlpnet_train_forward saved / output",
                 "tvr_mlpnet_train_forward mask bits", "tvr_gemm_tn_scaled scratch"):
        assert want in names, want


def test_render_scratch_header_is_the_only_kernel_owned_state(guards, tiny_dump, tiny_arrays, hyper_tiny):
    """tvr_render through the C-ABI with scratch of EXACTLY tvr_render_scratch_bytes + guard and outputs of exactly [n,3] / [n] + guard, 0xFF-poisoned: the call
    zeroes its own 256-byte header (words 0..3 public, 16 / 32 the kernels' tickets), leaves the guards alone, and gives the same picture as on fresh scratch."""
    from jittor_myc_nerfs_amd.autograd_ops import _stream_ptr
    L = guards
    lib = L.lib()
    m = make_model(tiny_arrays, hyper_tiny)
    rays = torch.tensor(tiny_dump["rays"], device="cuda")
    ref, _ = m.render_rays(rays, white_bg=True, N_samples=TINY["N_samples"])
    sc = m._ensure_scene()
    n, S = rays.shape[0], TINY["N_samples"]
    need = lib.tvr_render_scratch_bytes(sc, n, S)
    scratch = L.dev_bytes(need, "cuda", what="exact scratch")
    scratch.fill_(0xFF)
    rgb, depth = L.dev_empty((n, 3), torch.float32, "cuda", "exact rgb"), L.dev_empty((n,), torch.float32, "cuda", "exact depth")
    L.check(lib.tvr_render(sc, rays.data_ptr(), n, S, 1, None, float(m.rayMarch_weight_thres), rgb.data_ptr(), depth.data_ptr(), scratch.data_ptr(), need, None, None,
                           None, _stream_ptr(rays.device)), "tvr_render")
    torch.cuda.synchronize()
    assert torch.equal(rgb, ref)
    hdr = scratch[:256].view(torch.int32)
    assert int(hdr[2]) == 0 and int(hdr[0]) > 0                                  # no fault; the queue length
    assert lib.tvr_render(sc, rays.data_ptr(), n, S, 1, None, float(m.rayMarch_weight_thres), rgb.data_ptr(), depth.data_ptr(), scratch.data_ptr(), need - 1, None,
                          None, None, _stream_ptr(rays.device)) == -3            # one byte short: refused on the host
    assert L.check_guards() == [], L.check_guards()
