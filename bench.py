#!/usr/bin/env python3
"""bench.py — ray-samples/s of the TensoRF render path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one batch of synthetic rays resident in HBM:
  N = 1 : one 800x800 frame = 640 000 rays x 512 samples of scene A (TensorVMSplit 300^3) — BASELINE configs[1];
          ONE tvr_render call (march + shade + composite kernels).
  N > 1 : a batch of N such frames (N camera poses); the N*640 000 rays are cut into 4096-ray tiles dealt
          round-robin to the ranks (every rank renders 640 000 rays: weak scaling), then ONE all_gather of
          [rays,4] fp32 pixels (rgb+depth) over RCCL/xGMI returns all N frames to every rank (configs[2]).
value = nominal ray-samples/s = (rays x 512) / time, whole job (every ray counted with all 512 samples, masked or
terminated or not — SURVEY.md §8d).  Timed region: barrier + synchronize, K steps, synchronize + barrier; MAX over ranks.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

N_POSES = 8
TILE = 4096


def build_model(device, name="TensorVMSplit"):
    from jittor_myc_nerfs_amd import REFTensoRF, TensorVMSplit, synthetic
    A = synthetic.SCENE_A
    arrs = synthetic.make_scene_arrays(A["gridSize"], A["aabb"], ref=(name == "REFTensoRF"))
    H = synthetic.HYPER
    m = (REFTensoRF if name == "REFTensoRF" else TensorVMSplit)(arrs["aabb"], A["gridSize"], device, density_n_comp=[16] * 3, appearance_n_comp=[48] * 3, app_dim=27,
                      near_far=A["near_far"], shadingMode="MLP_Fea", alphaMask_thres=1e-4, density_shift=H["density_shift"],
                      distance_scale=H["distance_scale"], rayMarch_weight_thres=H["rayMarch_weight_thres"], pos_pe=6,
                      view_pe=2, fea_pe=2, featureC=128, step_ratio=A["step_ratio"], fea2denseAct=H["fea2denseAct"])
    m.load_arrays(arrs)
    return m, arrs, A


def frames(A):
    from jittor_myc_nerfs_amd import rays as R
    W, Hh = A["img_wh"]
    return [R.frame_rays(M, Hh, W, A["camera_angle_x"]) for M in R.sphere_poses(N_POSES, A["cam_radius"])]


def usable_cores():
    """Threads this process may really use: cgroup CPU quota if set, else the affinity mask, capped at the GPU box's
    per-GPU CPU share (16) so an over-subscribed pool does not distort the baseline."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("TVR_CPU_THREADS", "16"))))


def cpu_baseline(arrs, A, rays_cpu, budget_s=15.0, with_c=True):
    """Restated CPU path (oracle (a): the reference's op sequence on torch CPU, chunk 1024 as renderer.py:50), timed on
    a bounded strided sample of the same frame; plus the scalar-C oracle (b) with OpenMP for a second figure."""
    from jittor_myc_nerfs_amd import synthetic
    from oracle import c_oracle as CO, tensorf_oracle as TO
    cores = usable_cores()
    torch.set_num_threads(cores)
    hyper = dict(synthetic.HYPER, near_far=A["near_far"], step_ratio=A["step_ratio"])
    sc = TO.scene_from_arrays(arrs, **hyper)
    S = A["N_samples"]
    probe = rays_cpu[:: rays_cpu.shape[0] // 1024][:1024]
    t0 = time.perf_counter()
    TO.OctreeRender_trilinear_fast(probe, sc, chunk=1024, N_samples=S, white_bg=True)
    t_probe = time.perf_counter() - t0
    n = int(min(rays_cpu.shape[0], max(1024, (budget_s / max(t_probe, 1e-3)) * 1024)) // 1024 * 1024)
    stride = rays_cpu.shape[0] // n
    sample = rays_cpu[::stride][:n]
    t0 = time.perf_counter()
    TO.OctreeRender_trilinear_fast(sample, sc, chunk=1024, N_samples=S, white_bg=True)
    t = time.perf_counter() - t0
    out = {"value": n * S / t, "unit": "ray-samples/s", "cores": cores, "kind": "port",
           "sample": f"every {stride}th ray of pose 0: {n} rays x {S} samples in {t:.1f} s; restated CPU path = the reference's "
                     f"op sequence (tensorBase.py:476-536) on torch-CPU fp32, chunk 1024 (oracle/tensorf_oracle.py); "
                     f"Jittor itself cannot run here"}
    if not with_c:                     # the scalar-C restatement covers TensorVMSplit only
        return out
    co = CO.COracle(arrs, step=float(sc.stepSize), **hyper)
    n2 = min(n, 8192)
    sample2 = rays_cpu[:: rays_cpu.shape[0] // n2][:n2].numpy()
    t0 = time.perf_counter()
    co.render(sample2, S, white_bg=True, nthreads=cores)
    t2 = time.perf_counter() - t0
    out["scalar_c_value"] = n2 * S / t2
    out["scalar_c_note"] = f"oracle/tvr_oracle.c, OpenMP {cores} threads, {n2} rays in {t2:.1f} s"
    return out


def bench_ngp(args, world, rank, device):
    """BASELINE configs[4]: JNeRF Instant-NGP inference, one 800x800 frame per step through the fused frame path (tvr_ngp_render).  The
    path has no exchange step: for N > 1 every rank renders its own frames (replicas only)."""
    import math
    import torch.distributed as dist
    from jittor_myc_nerfs_amd import ngp, rays as R, synthetic
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("TVR_BENCH_BACKEND", "nccl")
        dist.init_process_group(backend, device_id=device) if backend == "nccl" else dist.init_process_group(backend)
    aabb_scale, W = 4, 800
    model = ngp.NGPNetworks(aabb_scale).to(device)
    sampler = ngp.DensityGridSampler(model, aabb_scale, rng=ngp.Pcg32(1337)).to(device)
    arrs = synthetic.make_ngp_scene_arrays(model.pos_encoder.offsets)
    ngp.load_scene_arrays(model, sampler, arrs)
    focal = 0.5 * W / math.tan(0.5 * 0.6911)
    poses = R.sphere_poses(8, 4.0)
    frames = [ngp.generate_rays(ngp.matrix_nerf2ngp(p), W, W, (focal, focal), device=device) for p in poses]
    for i in range(args.warmup):
        sampler.render_frame(*frames[i % len(frames)])

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
    marched = evaluated = 0
    k_ms = [0.0, 0.0]
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        st, pr = {}, {}
        sampler.render_frame(*frames[(args.warmup + i) % len(frames)], stats=st, profile=pr)
        marched += st["samples"]
        evaluated += st["evaluated"]
        k_ms[0] += pr["march_ms"]
        k_ms[1] += pr["render_ms"]
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    k_ms = [k / args.steps for k in k_ms]
    ev_per = evaluated / args.steps
    BYTES = 16 * 8 * 8 + 4 + 12                  # per evaluated sample: 16 levels x 8 corners x 8 B of table, its recorded t, 12 B of output share
    ach = BYTES * ev_per / (k_ms[1] * 1e-3) / 1e9
    result = {
        "metric": "ray_samples_per_sec", "value": marched * world / dt, "unit": "ray-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "JNeRF Instant-NGP inference (BASELINE configs[4]): 16-level hash grid 2^19 x 2, SH-16, MLPs 32-64-16 / 32-64-64-3, 5 x 128^3 "
                               "occupancy bitfield, 800x800 rays per GPU, up to 1024 steps per ray, dt = sqrt(3)/2048; fused frame path tvr_ngp_render; "
                               "N > 1: independent replicas (no exchange step on this path)",
                   "scene": "synthetic.make_ngp_scene_arrays, aabb_scale 4 (hollow-ball occupancy + 2 % speckle)", "rays_per_step": W * W * world,
                   "samples": "value counts the occupied steps the march produces (what the reference feeds its networks); the frame kernel "
                              "evaluates only those in front of the compositor's T < 1e-4 break"},
        "rays_per_sec": W * W * world * args.steps / dt,
        "effective": {"marched_samples_per_frame": marched / args.steps, "evaluated_samples_per_frame": ev_per},
        "kernel_ms": {"march": k_ms[0], "render": k_ms[1]},
        "roofline": {"kernel": "ngp_render_kernel<true>", "bound": "hbm", "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0,
                     "traffic": _ngp_traffic(), "algorithmic_bytes_per_launch": BYTES * ev_per, "ms": k_ms[1],
                     "note": f"{BYTES} B per evaluated sample; the 52 MB of tables sit in L2 / MALL, so the measured traffic is fabric traffic of random 128-B lines"},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import ngp_oracle as N
        o, d = frames[0]
        sel = torch.arange(0, W * W, 311, device=device)[:2048]                                 # a strided sample of pose 0
        on, dn = o[sel].cpu().numpy(), d[sel].cpu().numpy()
        levels = N.grid_levels(aabb_scale)
        a2 = dict(arrs)
        a2["density_grid_bitfield"], _ = N.update_bitfield(arrs["density_grid"])
        t1 = time.perf_counter()
        coords, _, numsteps, _, _ = N.sample(on, dn, a2["density_grid_bitfield"], aabb_scale, N.Pcg32(1337).state)
        out = N.network_c(levels, a2, coords)
        N.composite_c(out, coords, numsteps)
        tc = time.perf_counter() - t1
        result["cpu_baseline"] = {"value": coords.shape[0] / tc, "unit": "ray-samples/s", "cores": 1, "kind": "port",
                                  "sample": f"every 311th ray of pose 0: {len(on)} rays, {coords.shape[0]} samples in {tc:.1f} s; oracle/ngp_oracle.c "
                                            "(scalar C restatement: march + hash grid + SH + MLPs + compositing, every sample evaluated); the reference's CUDA/Jittor path cannot run here"}
    elif rank == 0:
        result["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


def _ngp_traffic():
    """FETCH_SIZE (x2, gfx950) of ngp_render_kernel from the committed PMC pass, bytes per launch."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r01_ngp_pmc.json")))
        for k, v in d.items():
            if "ngp_render_kernel" in k and "FETCH_SIZE" in v:
                return v["FETCH_SIZE"] * 1024.0 * 2.0
    except Exception:
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eps-T", type=float, default=None, help="early-termination threshold (default = weight thres 1e-4; 0 = exact)")
    ap.add_argument("--chunk", type=int, default=0, help="rays per tvr_render call (0 = the rank's whole batch in one call; 4096 = BASELINE "
                                                          "configs[3] / train.py batch size)")
    ap.add_argument("--alpha-mask", type=int, default=0, help="build an AlphaGridMask of this resolution with updateAlphaMask first "
                                                               "(the reference does so at iteration 2000/4000); 0 = none")
    ap.add_argument("--model", choices=["TensorVMSplit", "REFTensoRF", "NGPNetworks"], default="TensorVMSplit",
                    help="model_name (opt.py:44): TensorVMSplit is the BASELINE workload; REFTensoRF is the variant configs/Scar.txt trains; "
                         "NGPNetworks is the JNeRF Instant-NGP alt path (BASELINE configs[4]; replicas only for N > 1)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if args.gpus != 1 and world == 1:
            raise SystemExit(f"--gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py ...`")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the render path has no CPU fallback")
    dev_index = local_rank % torch.cuda.device_count()              # (rehearsals may put several ranks on one card)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    import torch.distributed as dist
    if args.model == "NGPNetworks":
        return bench_ngp(args, world, rank, device)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("TVR_BENCH_BACKEND", "nccl")       # "nccl" IS RCCL on ROCm; "gloo" only for 1-GPU rehearsals
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    from jittor_myc_nerfs_amd import _lib as L, shard_capacity, shard_indices
    import ctypes as C
    model, arrs, A = build_model(device, args.model)
    S = A["N_samples"]
    if args.alpha_mask > 0:
        model.updateAlphaMask((args.alpha_mask,) * 3)
    fr = frames(A)                                                  # 8 poses x [640000,6] on the host
    R1 = fr[0].shape[0]

    # per-step inputs, resident in HBM before the timed region.  Step s renders poses (s*N + r) % 8, r < N.
    n_patterns = N_POSES if world < N_POSES else 1
    cap = shard_capacity(world * R1, world, TILE)
    step_rays, step_idx = [], []
    for pat in range(n_patterns):
        batch = torch.cat([fr[(pat * world + r) % N_POSES] for r in range(world)]) if world > 1 else fr[pat]
        idx = shard_indices(batch.shape[0], rank, world, TILE)
        step_rays.append(batch[idx].contiguous().to(device))
        step_idx.append(idx)
    n_mine = step_rays[0].shape[0]
    all_idx = [shard_indices(world * R1, r, world, TILE).to(device) for r in range(world)] if world > 1 else None
    mine = torch.zeros((cap, 4), device=device)
    gathered = torch.empty((world * cap, 4), device=device) if world > 1 else None
    out_img = torch.empty((world * R1, 4), device=device) if world > 1 else None

    prof = C.c_void_p()
    L.check(L.lib().tvr_profile_create(max(args.steps, 1), C.byref(prof)), "tvr_profile_create")

    def step(s, profile=None, stats=None):
        rays = step_rays[s % n_patterns]
        if args.chunk > 0:                                          # renderer.py:16-25 chunk loop, one tvr_render per chunk, no host sync
            outs = [model.render_rays(rays[c0:c0 + args.chunk], white_bg=True, N_samples=S, eps_T=args.eps_T, stats=stats)
                    for c0 in range(0, rays.shape[0], args.chunk)]
            rgb, depth = torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])
        else:
            rgb, depth = model.render_rays(rays, white_bg=True, N_samples=S, eps_T=args.eps_T, stats=stats, profile=profile)
        if world > 1:
            mine[:n_mine, :3] = rgb
            mine[:n_mine, 3] = depth
            if dist.get_backend() == "gloo":                        # rehearsal path only
                parts = [torch.empty_like(mine) for _ in range(world)]
                dist.all_gather(parts, mine)
                gathered.copy_(torch.cat(parts))
            else:
                dist.all_gather_into_tensor(gathered, mine)
            for r in range(world):                                  # undo the tile interleave: index permutation only
                out_img.index_copy_(0, all_idx[r], gathered[r * cap:r * cap + all_idx[r].numel()])
            return out_img
        return rgb

    for s in range(args.warmup):
        step(s)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.steps):
        step(args.warmup + s, profile=prof)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    ms = (C.c_float * 3)()
    n_calls = L.lib().tvr_profile_read(prof, C.byref(ms))
    L.check(n_calls, "tvr_profile_read")
    k_ms = [ms[i] / max(n_calls, 1) for i in range(3)]
    if n_calls == 0:                                               # chunked mode records no per-kernel events
        k_ms = [0.0, 0.0, 0.0]

    # occupancy statistics of exactly the timed steps (untimed pass with counters on)
    stats = torch.zeros(8, dtype=torch.int64, device=device)
    for s in range(args.steps):
        step(args.warmup + s, stats=stats)
    torch.cuda.synchronize()
    st = stats.cpu().numpy().astype(np.float64) / max(args.steps, 1)      # per launch (this rank)
    m_eval, m_bbox, m_app = float(st[0]), float(st[1]), float(st[2])

    total_rays = world * R1
    value = total_rays * S * args.steps / dt
    has_mask = model.alphaMask is not None
    march_bytes = 40.0 * n_mine + 1152.0 * m_eval + (32.0 * m_bbox if has_mask else 0.0)
    shade_bytes = 3456.0 * m_app
    # HBM traffic per launch from the committed rocprofv3 PMC passes (FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950
    # correction + WRITE_SIZE, KB -> B); only valid for the workload it was collected on (this one)
    pmc = {}
    pmc_path = os.path.join(ROOT, "profiles", "pmc_summary.json")
    if os.path.exists(pmc_path) and args.model == "TensorVMSplit" and args.alpha_mask == 0:
        try:
            pmc = json.load(open(pmc_path))
        except Exception:
            pmc = {}
    t_march, t_shade = k_ms[0] * 1e-3, k_ms[1] * 1e-3
    FLOP_APP = 8.0e4            # algorithmic FLOP per appearance sample (basis 7 776 + MLP 71 936 + PE)
    FLOP_APP_EXEC = 3 * 2 * (32 * 144 + 128 * 160 + 128 * 128 + 32 * 128)   # executed on the matrix cores: 3 fp16 products, padded
    if args.model == "REFTensoRF":                                         # + four 144 -> {3,3,1,1} heads, 151-input layer 1
        FLOP_APP += 2 * 8 * 144 + 2 * 128
        FLOP_APP_EXEC += 3 * 2 * 32 * 144
    roof_march = {"kernel": "march_kernel<false>", "bound": "hbm", "achieved": march_bytes / t_march / 1e9 if t_march > 0 else None,
                  "peak": 8000.0, "unit": "GB/s", "frac": march_bytes / t_march / 1e9 / 8000.0 if t_march > 0 else None,
                  "traffic": pmc.get("march_hbm_bytes_per_launch"), "algorithmic_bytes_per_launch": march_bytes, "ms": k_ms[0],
                  "note": "40 B/ray + 1152 B per density sample actually evaluated (+32 B per alpha-mask lookup); the 17 MB of density "
                          "factors are L2/Infinity-Cache resident (frac > 1 against HBM): the kernel runs at the L1 (TA) rate"}
    roof_shade = {"kernel": "shade_kernel<0,0,%s>" % ("true" if args.model == "REFTensoRF" else "false"), "bound": "mfma", "achieved": FLOP_APP * m_app / t_shade / 1e12 if t_shade > 0 else None,
                  "peak": 2500.0, "unit": "TFLOP/s", "frac": FLOP_APP * m_app / t_shade / 1e12 / 2500.0 if t_shade > 0 else None,
                  "traffic": pmc.get("shade_hbm_bytes_per_launch"), "algorithmic_flops_per_launch": FLOP_APP * m_app, "ms": k_ms[1],
                  "executed_mfma_TFLOPs": FLOP_APP_EXEC * m_app / t_shade / 1e12 if t_shade > 0 else None,
                  "gather_algorithmic_GBps": shade_bytes / t_shade / 1e9 if t_shade > 0 else None,
                  "note": f"{FLOP_APP / 1e3:.1f} kFLOP per appearance sample (algorithmic, fp32 semantics) against the dense f16 MFMA peak; the kernel "
                          "executes 3 fp16 products per fp32 product (hi/lo split, fp32-class accuracy) on padded tiles = "
                          f"{FLOP_APP_EXEC} FLOP per sample, and gathers 3456 B per sample through L1"}
    dominant = roof_shade if k_ms[1] >= k_ms[0] else roof_march
    result = {
        "metric": "ray_samples_per_sec", "value": value, "unit": "ray-samples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("TensorVMSplit 300^3 (16/48 comps, MLP_Fea 150-128-128-3)" if args.model == "TensorVMSplit" else
                                "REFTensoRF 300^3 (16/48 comps, 4 heads on h, MLP_Fea_Ref 151-128-128-3)") +
                               ", 800x800 rays x 512 samples/ray per GPU "
                               "(BASELINE configs[1]; N>1: configs[2] as N frames, 4096-ray tiles round-robin, one RCCL all_gather "
                               "of [rays,4] fp32)", "scene": "synthetic scene A (SURVEY 8d), no alpha mask, white_bg",
                   "rays_per_step": total_rays, "samples_per_ray": S,
                   "eps_T": float(model.rayMarch_weight_thres) if args.eps_T is None else args.eps_T, "tile": TILE,
                   "chunk": args.chunk if args.chunk > 0 else n_mine, "alpha_mask": args.alpha_mask},
        "rays_per_sec": total_rays * args.steps / dt,
        "effective": {"density_samples_evaluated_per_sec": m_eval * world * args.steps / dt,
                      "appearance_samples_per_sec": m_app * world * args.steps / dt,
                      "frac_samples_evaluated": m_eval / (n_mine * S), "frac_samples_in_box": m_bbox / (n_mine * S),
                      "app_samples_per_ray": m_app / n_mine},
        "kernel_ms": {"march": k_ms[0], "shade": k_ms[1], "composite": k_ms[2], "calls": n_calls},
        "roofline": dominant,
        "roofline_all": {"march": roof_march, "shade": roof_shade},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(arrs, A, fr[0], with_c=(args.model == "TensorVMSplit"))
    elif rank == 0:
        result["cpu_baseline"] = None
    L.lib().tvr_profile_destroy(prof)
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
