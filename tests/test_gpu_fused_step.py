"""GPU (-m gpu): the training step without a host read (tvr_train_forward / tvr_train_backward; autograd_ops._FusedStepFn) —
  * the step is a fixed launch sequence: captured in a hipGraph (forward, backward AND the fused Adam update), and a replay equals the eager step bit for bit;
  * two eager steps on the same batch are bit-identical (fixed-order compositing sums and reductions; torch's index_add in the eager chain is not);
  * a batch whose appearance samples exceed the workspace is FLAGGED (check_training_faults() -> 'overflow', capacity doubled), never silently truncated;
  * static step == eager chain to rounding (the eager chain composites with index_add)."""
import numpy as np
import pytest
import torch

from conftest import TINY, make_model

pytestmark = pytest.mark.gpu


def _batch(tiny_dump, n_rep=8, seed=3):
    rays = np.concatenate([tiny_dump["rays"]] * n_rep).copy()
    rays[:, :3] += 0.01 * np.random.default_rng(seed).standard_normal((rays.shape[0], 3)).astype(np.float32)
    return torch.tensor(rays, device="cuda")


def _params(m):
    return [p for g in m.get_optparam_groups(0.02, 0.001) for p in g["params"]]


def _step_fn(m, rays, target, jitter, opt):
    def step():
        opt.zero_grad(set_to_none=False)
        rgb, _ = m.render_rays_autograd(rays, white_bg=True, N_samples=TINY["N_samples"], jitter=jitter)
        loss = torch.mean((rgb - target) ** 2)
        if hasattr(m, "penalty"):
            loss = loss + 0.5 * m.penalty
        loss.backward()
        opt.step()
        return loss
    return step


@pytest.mark.parametrize("ref", [False, True])
def test_training_step_is_graph_capturable_and_replay_equals_eager(ref, tiny_dump, tiny_arrays, tiny_ref_arrays, hyper_tiny):
    arrs = tiny_ref_arrays if ref else tiny_arrays
    rays = _batch(tiny_dump)
    target = torch.rand((rays.shape[0], 3), device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    jitter = torch.rand(rays.shape[0], device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))

    def fresh():
        m = make_model(arrs, hyper_tiny)
        opt = torch.optim.Adam(m.get_optparam_groups(0.02, 0.001), betas=(0.9, 0.99), capturable=True, foreach=True)
        return m, opt

    # eager: three steps
    m1, o1 = fresh()
    s1 = _step_fn(m1, rays, target, jitter, o1)
    losses_e = [float(s1().detach()) for _ in range(3)]
    assert m1.check_training_faults() is None and m1._train_buf is not None
    # graph: one warm-up step on a side stream (allocations, buffers, Adam state), then capture ONE step and replay it twice -> also three steps
    m2, o2 = fresh()
    s2 = _step_fn(m2, rays, target, jitter, o2)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        l0 = s2()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        lg = s2()
    losses_g = [float(l0.detach())]
    g.replay(); torch.cuda.synchronize()              # (capture itself does not execute: this is step 2)
    losses_g.append(float(lg.detach()))
    g.replay(); torch.cuda.synchronize()
    losses_g.append(float(lg.detach()))
    assert m2.check_training_faults() is None
    assert losses_g == losses_e, (losses_g, losses_e)
    for a, b in zip(_params(m1), _params(m2)):
        assert torch.equal(a, b), "a parameter differs between three eager steps and warm-up + two graph replays"
    assert losses_e[2] < losses_e[0]


def test_static_step_is_bit_reproducible_and_agrees_with_the_eager_chain(tiny_dump, tiny_arrays, hyper_tiny):
    rays = _batch(tiny_dump, 16)
    cw = torch.randn((rays.shape[0], 3), device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))

    def grads(static):
        m = make_model(tiny_arrays, hyper_tiny)
        m.static_training = static
        rgb, _ = m.render_rays_autograd(rays, white_bg=True, N_samples=TINY["N_samples"])
        (rgb * cw).sum().backward()
        return rgb.detach(), [p.grad.clone() for p in _params(m)]
    rgb_a, ga = grads(True)
    rgb_b, gb = grads(True)
    assert torch.equal(rgb_a, rgb_b) and all(torch.equal(x, y) for x, y in zip(ga, gb)), "two static steps on the same batch differ"
    rgb_e, ge = grads(False)
    assert float((rgb_a - rgb_e).abs().max()) < 2e-6
    for x, y in zip(ga, ge):
        assert float((x - y).abs().max()) <= 2e-5 * max(float(y.abs().max()), 1e-6) + 1e-9


def test_workspace_overflow_is_flagged_not_truncated(tiny_dump, tiny_arrays, hyper_tiny):
    rays = _batch(tiny_dump, 64)                                       # 4096 rays
    m = make_model(tiny_arrays, hyper_tiny)
    m.train_app_samples_per_ray = 1                                    # capacity max(4096, 1 x 4096) = 4096 entries; the batch shades more
    rgb, _ = m.render_rays_autograd(rays, white_bg=True, N_samples=TINY["N_samples"])
    rgb.sum().backward()
    lay_cap = m._train_buf["cap"]
    assert m.check_training_faults() == "overflow" and m.train_app_samples_per_ray == 2 and m._train_buf is None
    m.train_app_samples_per_ray = 192
    rgb2, _ = m.render_rays_autograd(rays, white_bg=True, N_samples=TINY["N_samples"])
    rgb2.sum().backward()
    assert m.check_training_faults() is None and m._train_buf["cap"] > lay_cap
    with torch.no_grad():
        ref, _ = m.render_rays(rays, white_bg=True, N_samples=TINY["N_samples"])
    assert float((rgb2.detach() - ref).abs().max()) < 2e-6
