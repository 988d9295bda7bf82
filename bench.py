#!/usr/bin/env python3
"""bench.py — ray-samples/s of the TensoRF render path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one batch of synthetic rays resident in HBM:
  N = 1 : one 800x800 frame = 640 000 rays x 512 samples of scene A (TensorVMSplit 300^3) — BASELINE configs[1];
          ONE tvr_render call (march + shade + composite kernels).
  N > 1 (default, --scaling strong): BASELINE configs[2] as written — the SAME ONE 640 000-ray frame, cut into 4096-ray tiles dealt
          round-robin to the ranks (rank r renders tiles r, r+N, ...: 80 000 rays each at N = 8) straight into its send buffer, then ONE
          all_gather of [4 cap] fp32 (rgb block + depth block) over RCCL/xGMI returns the frame to every rank and two strided copies undo the interleave.  Total work is
          fixed, so the line says "scaling": "strong" at every N (N = 1 included) and SURVEY 8e's t_1 / (N t_N) is value(N) / (N value(1)).
          Rank 0 also times the whole frame alone (outside the timed region) and reports it as `strong_split.t1_ms`.
  N > 1, --scaling weak (on request): a batch of N such frames (N camera poses), every rank renders 640 000 rays.
  --emulate-world N (single GPU): time rank 0's share of an N-way strong split without the exchange (what a rank's
          kernels cost at that size: persistent-kernel fill, LDS image reload).
value = nominal ray-samples/s = (rays x 512) / time, whole job (every ray counted with all 512 samples, masked or
terminated or not — SURVEY.md §8d).  Timed region: barrier + synchronize, K steps, synchronize + barrier; MAX over ranks.

roofline.traffic is measured by THIS run: after the timed region, rank 0 (N = 1) re-runs a 2-step bench under
`rocprofv3 --pmc` in child processes, one pass per counter set (FETCH_SIZE; WRITE_SIZE; SQ/GRBM sets), as
MI355X_MICROARCH.md's HBM section prescribes (FETCH_SIZE doubled on gfx950).  If the profiler is unavailable the field is
null and `traffic_source` says why — never a constant read from a committed file.
"""
import argparse
import json
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

N_POSES = 8
TILE = 512         # rays per shard tile: round 4 (scripts/shard_balance.py): the slowest of 8 ranks carries 1.020 x the mean march work with 4096-ray tiles, 1.0045 x with 512
SHADE_LOADS_PER_TILE = 114      # tvr_shade.hip, wave-level global loads per 32-entry tile as PMC counts them (SQ_INSTS_VMEM_RD, rounds 3 and 4): 108 taps + 3 basis fragments (lo parts of k-steps 6..8) + 3 entry / direction; fallback only, the live counter is used when the PMC pass ran


def build_model(device, name="TensorVMSplit"):
    from jittor_myc_nerfs_amd import REFTensoRF, TensorVMSplit, synthetic
    A = synthetic.SCENE_A
    pe = int(os.environ.get("TVR_PE", "2"))               # encoding frequencies (view_pe = fea_pe): 6 = TensorBase.__init__'s own default (scripts/train_step_timing.py)
    arrs = synthetic.make_scene_arrays(A["gridSize"], A["aabb"], ref=(name == "REFTensoRF")) if pe == 2 else \
        synthetic.make_scene_arrays(A["gridSize"], A["aabb"], view_pe=pe, fea_pe=pe)
    H = synthetic.HYPER
    m = (REFTensoRF if name == "REFTensoRF" else TensorVMSplit)(arrs["aabb"], A["gridSize"], device, density_n_comp=[16] * 3, appearance_n_comp=[48] * 3, app_dim=27,
                      near_far=A["near_far"], shadingMode="MLP_Fea", alphaMask_thres=1e-4, density_shift=H["density_shift"],
                      distance_scale=H["distance_scale"], rayMarch_weight_thres=H["rayMarch_weight_thres"], pos_pe=6,
                      view_pe=pe, fea_pe=pe, featureC=128, step_ratio=A["step_ratio"], fea2denseAct=H["fea2denseAct"])
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


_JSON_FD = None


def quiet_stdout():
    """The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a five-line version banner on stdout at communicator
    creation): point fd 1 at stderr for the run and keep the real stdout for the line."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def emit(result):
    line = (json.dumps(result) + "\n").encode()
    sys.stdout.flush()
    if _JSON_FD is None:
        sys.stdout.buffer.write(line)
        sys.stdout.flush()
    else:
        os.write(_JSON_FD, line)


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
    # roofline.traffic, live, as on the main path: rocprofv3 --pmc child passes of this same command after the timed region (rank 0, N = 1)
    pmc, pmc_source = {}, "not collected (--pmc off or N > 1)"
    if args.pmc == "auto" and world == 1 and rank == 0:
        pmc, pmc_source = collect_pmc(["--model", "NGPNetworks"], passes=PMC_PASSES[:3])
    pr_ = _pmc_kernel(pmc, "ngp_render_kernel")
    traffic = (2.0 * pr_["FETCH_SIZE"] + pr_["WRITE_SIZE"]) * 1024.0 if ("FETCH_SIZE" in pr_ and "WRITE_SIZE" in pr_) else None
    result = {
        "metric": "ray_samples_per_sec", "value": marched * world / dt, "unit": "ray-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 (emulated: 3 x f16-split MFMA products, fp32 accumulate; fp32 VALU elsewhere)", "data": "synthetic",
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
                     "traffic": traffic, "traffic_source": pmc_source, "algorithmic_bytes_per_launch": BYTES * ev_per, "ms": k_ms[1],
                     "clock_GHz_pmc_pass": (pr_["GRBM_GUI_ACTIVE"] / 8.0 / pr_["_dur_s"] / 1e9) if (pr_.get("GRBM_GUI_ACTIVE") and pr_.get("_dur_s")) else None,
                     "mfma_busy_frac": (pr_["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (pr_["GRBM_GUI_ACTIVE"] / 8.0)) if pr_.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in pr_ else None,
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
        emit(result)
    if world > 1:
        dist.destroy_process_group()


PMC_PASSES = (("FETCH_SIZE",), ("WRITE_SIZE",),
              ("GRBM_GUI_ACTIVE", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES",
               "SQ_VALU_MFMA_COEXEC_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES"),
              ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "TCC_HIT_sum", "TCC_MISS_sum"),
              ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM"))


def collect_pmc(extra_args, budget_s=420.0, passes=None):
    """One rocprofv3 --pmc pass per counter set over `bench.py --steps 2 --warmup 1` in a child process (program directly behind `--`).
    Returns ({kernel short name: {counter: mean per dispatch}}, source string)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {}, "rocprofv3 not found: traffic not measured"
    out, t_start, done = {}, time.time(), []
    env = dict(os.environ, TMPDIR="/tmp")
    for counters in (passes or PMC_PASSES):
        if time.time() - t_start > budget_s:
            break
        d = tempfile.mkdtemp(prefix="tvr_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
               "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--pmc", "off", *extra_args]
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                shutil.rmtree(d, ignore_errors=True)
                continue
            acc = {}
            for f in files:
                for row in csv.DictReader(open(f)):
                    k = row["Kernel_Name"].split("(")[0].replace("void ", "")
                    acc.setdefault(k, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
                    if row["Counter_Name"] == "GRBM_GUI_ACTIVE" and row.get("End_Timestamp"):       # this pass's own kernel duration, for its own clock
                        acc[k].setdefault("_dur_s", []).append((float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) * 1e-9)
            for k, cs in acc.items():
                for c, v in cs.items():
                    out.setdefault(k, {})[c] = sum(v) / len(v)
            done.append("+".join(counters))
        except Exception:
            pass
        shutil.rmtree(d, ignore_errors=True)
    if not done:
        return {}, "rocprofv3 --pmc child passes failed on this box: traffic not measured"
    return out, ("live: rocprofv3 --pmc child passes of this bench run (2 steps each; mean per dispatch): " + "; ".join(done) +
                 "; traffic = 2 x FETCH_SIZE + WRITE_SIZE (KB -> B; gfx950 tallies 128-B reads at 64 B, MI355X_MICROARCH.md HBM section)")


def other_configs(budget_s=240.0):
    import subprocess
    out, t0 = {}, time.time()
    for key, extra in (("configs[3]: TensorVMSplit 300^3, 800x800 x 512 as 157 tvr_render calls of 4096 rays", ["--chunk", "4096"]),
                       ("configs[3] with two calls in flight: the same 157 calls through render.FrameStream (call k on stream k % 2)", ["--chunk", "4096", "--chunk-stream"]),
                       ("configs[4]: JNeRF Instant-NGP inference, 800x800, fused frame path", ["--model", "NGPNetworks"])):
        if time.time() - t0 > budget_s:
            out[key] = "skipped: time budget"
            continue
        cmd = [sys.executable, os.path.abspath(__file__), "--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--pmc", "off", *extra]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=200, env=dict(os.environ))
            d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
            out[key] = {k: d[k] for k in ("value", "unit", "ms_per_step", "rays_per_sec", "steps", "kernel_ms", "roofline") if k in d}
            if not any(v for k, v in out[key].get("kernel_ms", {}).items() if k != "calls"):
                out[key].pop("kernel_ms", None)                       # chunked calls are not timed per kernel
            out[key]["command"] = "python bench.py " + " ".join(cmd[2:])
        except Exception as e:                                      # never let an extra line break the contract line
            out[key] = f"failed: {type(e).__name__}"
    # the training step of the three models (train.py:225-261 at the reference's settings: 4096 rays, nSamples = 1039 at 300^3, MSE +
    # regularisers + fused Adam), timed by the scripts under scripts/ in child processes
    import re
    root = os.path.dirname(os.path.abspath(__file__))
    for key, script, env in (("training step: TensorVMSplit 300^3, 4096 rays x 1039 samples", "train_step_timing.py", {"TVR_MODEL": "TensorVMSplit"}),
                             ("training step: REFTensoRF 300^3 (configs/Scar.txt), 4096 rays x 1039 samples", "train_step_timing.py", {"TVR_MODEL": "REFTensoRF"}),
                             # round 6: TensorBase.__init__'s own encoding frequencies (tensorBase.py:141-145: view_pe = fea_pe = 6, 390 MLP inputs) through the fused step
                             ("training step: TensorVMSplit 300^3 with view_pe = fea_pe = 6, 4096 rays x 1039 samples", "train_step_timing.py", {"TVR_MODEL": "TensorVMSplit", "TVR_PE": "6"}),
                             ("training step: NerfPlusPlus 300^3 (configs/Scarf.txt), 4096 rays x (1039 + 512 background) samples", "npp_train_step_timing.py", {})):
        if time.time() - t0 > budget_s:
            out[key] = "skipped: time budget"
            continue
        try:
            # Round 6 (review item 7): the eager static step is paced by the HOST (about 60 launches of 5 - 100 us kernels per step: 3.4 ms on a quiet host, 4.2 on the
            # driver's busy one); the same step captured once and replayed as ONE hipGraph (training.make_graphed_step's way; TVR_GRAPH=1) is paced by the GPU and is what a
            # reconstruction loop that replays gets — both are printed, the graph's first
            res = {}
            for tag, genv in (("graph", {"TVR_GRAPH": "1"}), ("eager", {"TVR_GRAPH": "0"})):
                if script != "train_step_timing.py" and tag == "graph":
                    continue
                r = subprocess.run([sys.executable, os.path.join(root, "scripts", script)], capture_output=True, text=True, timeout=120, env=dict(os.environ, **env, **genv))
                mt = re.search(r"train step[^:]*: *([0-9.]+) ms", r.stdout)
                res[tag] = float(mt.group(1))
            ms = res.get("graph", res["eager"])
            out[key] = {"ms_per_step": ms, "iterations_per_sec": 1e3 / ms, "rays_per_sec": 4096e3 / ms,
                        "ms_per_step_whole_step_hipgraph_replay": res.get("graph"), "ms_per_step_eager_host_paced": res["eager"],
                        "command": " ".join(f"{k}={v}" for k, v in env.items()) + (" " if env else "") + ("TVR_GRAPH=1 " if "graph" in res else "") + "python scripts/" + script}
        except Exception as e:
            out[key] = f"failed: {type(e).__name__}"
    # NerfPlusPlus inference: the background network's kernel (512 samples per ray: the bulk of a NerfPlusPlus frame) with its roofline (scripts/npp_roofline.py)
    key = "NerfPlusPlus inference (configs/Scarf.txt): background network kernel, 65 536 rays x 512 samples per launch"
    if time.time() - t0 > budget_s:
        out[key] = "skipped: time budget"
    else:
        try:
            r = subprocess.run([sys.executable, os.path.join(root, "scripts", "npp_roofline.py")], capture_output=True, text=True, timeout=120, env=dict(os.environ))
            d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
            out[key] = {"ms_per_launch": d["ms_per_launch"], "G_samples_per_s": d["G_samples_per_s"], "ms_per_800x800_frame_of_this_kernel": d["ms_per_800x800_frame_of_this_kernel"],
                        "roofline": d, "command": "python scripts/npp_roofline.py"}
        except Exception as e:
            out[key] = f"failed: {type(e).__name__}"
    return out


def _pmc_kernel(pmc, prefix):
    for k, v in pmc.items():
        if k.startswith(prefix):
            return v
    return {}


def frame_stream(model, step_rays, S, eps_T, frames_n=24):
    """Throughput of a STREAM of frames with two in flight (render.FrameStream: frame k on stream k % 2, own scratch and outputs), behind the timed region.
    Informational — `value` is the serial loop's, one frame at a time, whose kernels can be timed one by one."""
    from jittor_myc_nerfs_amd import FrameStream
    fs = FrameStream(model, white_bg=True, N_samples=S, eps_T=eps_T)
    for s in range(4):
        fs.submit(step_rays[s % len(step_rays)])
    fs.flush()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(frames_n):
        fs.submit(step_rays[s % len(step_rays)])
    last = fs.flush()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / frames_n
    ref = model.render_rays(step_rays[(frames_n - 1) % len(step_rays)], white_bg=True, N_samples=S, eps_T=eps_T)
    return {"ms_per_frame": dt * 1e3, "ray_samples_per_sec": step_rays[0].shape[0] * S / dt, "frames": frames_n,
            "last_frame_equals_serial_render": bool(torch.equal(last[0], ref[0]) and torch.equal(last[1], ref[1])),
            "note": "two frames in flight on two HIP streams (FrameStream): the persistent kernels of one frame fill the CUs the other frame's kernels leave as they drain; "
                    "not the headline — a frame's latency is not shortened and its kernels cannot be timed one by one"}


def arith_modes(model, step_rays, S, eps_T, m_app, steps=8):
    """The opt-in arithmetics of the appearance network (include/tvr.h, tvr_scene_set_arith) on the bench frame, behind the timed region: kernel times
    (HIP events of the library's own profile, `steps` frames over the poses) and the picture's distance from the default (fp32-class) mode's on ALL rays of
    every pose.  Informational: `value` is the default mode's."""
    import ctypes as C
    from jittor_myc_nerfs_amd import _lib as L
    out, ref = {}, None
    try:
        for mode in ("f32", "f16act", "f16"):
            model.mlp_arith = mode
            prof = C.c_void_p()
            L.check(L.lib().tvr_profile_create(steps, C.byref(prof)), "tvr_profile_create")
            model.render_rays(step_rays[0], white_bg=True, N_samples=S, eps_T=eps_T)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for s in range(steps):
                model.render_rays(step_rays[s % len(step_rays)], white_bg=True, N_samples=S, eps_T=eps_T, profile=prof)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            ms = (C.c_float * 3)()
            n = max(L.lib().tvr_profile_read(prof, C.byref(ms)), 1)
            L.lib().tvr_profile_destroy(prof)
            pics = [model.render_rays(r, white_bg=True, N_samples=S, eps_T=eps_T)[0] for r in step_rays]
            o = {"ms_per_frame": dt * 1e3, "kernel_ms": {"march": ms[0] / n, "shade": ms[1] / n, "composite": ms[2] / n},
                 "ray_samples_per_sec": step_rays[0].shape[0] * S / dt,
                 # the same algorithmic 80 kFLOP per appearance sample against the same dense-f16 peak as roofline.shade
                 "shade_frac_of_dense_f16_peak": (8.0e4 * m_app / (ms[1] / n * 1e-3) / 1e12 / 2500.0) if ms[1] > 0 and m_app > 0 else None,
                 # the gate (tvr_scene_validate_arith): the mode the kernels really ran in, and what the gate measured on its probe rays before letting it run
                 "in_effect": model.arith_in_effect, "gate_max_diff_on_probe_rays": (model.arith_max_diff if mode != "f32" else None), "gate_tolerance": model.mlp_arith_tol}
            if ref is None:
                ref = pics
            else:
                d = [(a - b).abs() for a, b in zip(pics, ref)]
                o["rgb_Linf_vs_f32_mode"] = max(float(x.max()) for x in d)
                o["rgb_mean_abs_vs_f32_mode"] = float(sum(x.mean() for x in d) / len(d))
                o["rays_compared"] = int(sum(x.shape[0] for x in d))
                o["nonfinite_pixels"] = int(sum((~torch.isfinite(a)).sum() for a in pics))
            out[mode] = o
    finally:
        model.mlp_arith = "f32"
    out["note"] = ("products per k-step of basis / layer 1 / layer 2: f32 = Wlo*xhi + Whi*xlo + Whi*xhi (default, the headline); f16act = Wlo*xhi + Whi*xhi (activations rounded to "
                   "fp16, nearest even); f16 = Whi*xhi.  north_star's parity bar: RGB L-inf 1e-3.  A reduced mode runs only after the library has measured it on this scene's own "
                   "rays against the default mode (`in_effect`; refused beyond `gate_tolerance`)")
    return out


def main():
    global TILE
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
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                    help="N > 1: strong (default) = ONE frame split over the ranks (BASELINE configs[2]; SURVEY 8e's metric); weak = N frames per step "
                         "(per-GPU work fixed)")
    ap.add_argument("--emulate-world", type=int, default=0, help="single GPU: render only rank 0's share of an N-way strong split (no exchange)")
    ap.add_argument("--check", action="store_true", help="N > 1: every rank also renders the whole batch alone and compares the gathered pixels bit for bit")
    ap.add_argument("--one-rank-exchange", action="store_true", help="N = 1 rehearsal of the N > 1 step on the RCCL backend: a one-rank `nccl` process group, the send "
                                                                      "buffer, all_gather_into_tensor and the un-permute, with --check (the only way one card can run the nccl branch)")
    ap.add_argument("--tile", type=int, default=TILE, help="rays per shard tile (tiles are dealt round-robin to the ranks)")
    ap.add_argument("--no-pipeline", action="store_true", help="split frames (N > 1, --emulate-world, --one-rank-exchange): plain calls per step — four launches, exchange on "
                                                               "the compute stream — instead of ShardedFramePipeline (hipGraph replay + exchange on a side stream)")
    ap.add_argument("--graph", action="store_true", help="pipeline WITH the per-rank render replayed as a hipGraph (measured: no gain — the four plain launches already "
                                                         "run without gaps, gpurun_out/r4m; kept as an A/B switch)")
    ap.add_argument("--no-overlap-exchange", action="store_true", help="emulation: leave the device-side exchange out of the timed loop (A/B)")
    ap.add_argument("--arith", choices=["f32", "f16act", "f16"], default="f32",
                    help="arithmetic of the appearance network's matrix products (tvr_scene_set_arith): f32 = three fp16 products per fp32 product, fp32-class — the headline; "
                         "f16act / f16 = the opt-in reduced modes (two / one product).  The default run reports all three under `arith_modes`")
    ap.add_argument("--autotune", action="store_true", help="decide pieces / one launch set on this card in the warm-up (model.autotune_render_pieces) instead of the library's default (pieces)")
    ap.add_argument("--two-shares-in-flight", action="store_true", help="split frames: two shares in flight on two render streams (ShardedFramePipeline(two_in_flight=True), round 6: "
                                                                       "-3 %% in the local emulation, not reproducible through RCCL's own stream — off by default, render.py)")
    ap.add_argument("--pieces", type=int, default=None, help="rays per piece of a tvr_render call rendered in pieces on two library-owned streams (include/tvr.h, PIECES): "
                                                              "default = the library's (30 720); 0 = one launch set per call, as before round 6")
    ap.add_argument("--chunk-stream", action="store_true", help="with --chunk: the chunk calls go through render.FrameStream (two calls in flight on two streams) "
                                                                "instead of one after the other")
    ap.add_argument("--no-extras", action="store_true", help="N = 1 default workload: do not append the BASELINE configs[3] / configs[4] lines (child runs of this script)")
    ap.add_argument("--pmc", choices=["auto", "off"], default="auto", help="auto: rank 0 at N = 1 measures roofline.traffic with rocprofv3 --pmc child passes")
    ap.add_argument("--img", type=int, default=int(os.environ.get("TVR_BENCH_IMG", "800")), help="frame edge in pixels (800 = the BASELINE workload; "
                                                                                              "smaller only for rehearsals / tests)")
    args = ap.parse_args()
    quiet_stdout()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or args.one_rank_exchange:
        # A rank of a split frame drives FOUR to FIVE streams (the caller's, the exchange's side stream, RCCL's own, the two piece streams of a large share) and HIP maps
        # streams onto four hardware queues by default: two of them then share a queue and serialise.  Eight queues (a ROCclr setting, read when the runtime starts —
        # nothing has touched the GPU yet): the 8-way share through a one-member RCCL group 2.43 -> 2.41 ms, the 4-way share 5.09 -> 5.04 (profiles/r06_share_in_flight_ab.txt, block 7).
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    if world != args.gpus:
        if args.gpus != 1 and world == 1:
            raise SystemExit(f"--gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py ...`")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if torch.cuda.device_count() == 0:                              # (counting devices does not initialise the GPU)
        raise SystemExit("bench.py needs an MI355X: the render path has no CPU fallback")
    pmc, pmc_source = {}, "not collected (--pmc off, N > 1, or a non-default workload)"
    default_workload = args.model == "TensorVMSplit" and args.chunk == 0 and args.emulate_world == 0 and args.img == 800 and not args.one_rank_exchange
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the render path has no CPU fallback")
    dev_index = local_rank % torch.cuda.device_count()              # (rehearsals may put several ranks on one card)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    import torch.distributed as dist
    if args.model == "NGPNetworks":
        return bench_ngp(args, world, rank, device)
    dist_on = world > 1 or args.one_rank_exchange                   # the exchange path runs (one-rank rehearsal: world stays 1)
    if dist_on:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:
            for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", str(36100 + os.getpid() % 2000)), ("RANK", "0"), ("WORLD_SIZE", "1")):
                os.environ.setdefault(k, v)
        backend = os.environ.get("TVR_BENCH_BACKEND", "nccl")       # "nccl" IS RCCL on ROCm; "gloo" only for 1-GPU rehearsals
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    from jittor_myc_nerfs_amd import _lib as L, shard_capacity, shard_indices, shard_send_views, shard_unpermute, ShardedFramePipeline
    TILE = args.tile
    import ctypes as C
    model, arrs, A = build_model(device, args.model)
    model.mlp_arith = args.arith
    if args.pieces is not None:
        model.render_piece_rays = args.pieces
    n_prod = {"f32": 3, "f16act": 2, "f16": 1}[args.arith]
    S = A["N_samples"]
    if args.alpha_mask > 0:
        model.updateAlphaMask((args.alpha_mask,) * 3)
    if args.img != 800:
        A = dict(A, img_wh=(args.img, args.img))
    fr = frames(A)                                                  # 8 poses x [img*img,6] on the host
    R1 = fr[0].shape[0]
    strong = args.scaling == "strong" and dist_on
    split = world if world > 1 else max(args.emulate_world, 1)      # ways the step's batch is cut (emulation: only rank 0's share is rendered)
    frames_per_step = 1 if (strong or world == 1) else world
    R_step = frames_per_step * R1                                   # rays per step, whole job

    # per-step inputs, resident in HBM before the timed region.  Weak: step s renders poses (s*N + r) % 8, r < N.  Strong: pose s % 8.
    n_patterns = N_POSES if frames_per_step < N_POSES else 1
    cap = shard_capacity(R_step, split, TILE)
    step_rays, full_rays = [], []
    for pat in range(n_patterns):
        batch = torch.cat([fr[(pat * frames_per_step + r) % N_POSES] for r in range(frames_per_step)]) if frames_per_step > 1 else fr[pat % N_POSES]
        idx = shard_indices(R_step, rank, split, TILE)
        step_rays.append(batch[idx].contiguous().to(device))
        if dist_on and (args.check or strong):
            full_rays.append(batch.to(device))
    n_mine = step_rays[0].shape[0]
    # the send buffer of the exchange: [rgb block 3 cap | depth block cap] fp32; the render kernels write straight into views of it
    mine = torch.zeros((4 * cap,), device=device)
    send_views = shard_send_views(mine, cap, n_mine) if dist_on else None
    gathered = torch.empty((world * 4 * cap,), device=device) if dist_on else None
    gloo = dist_on and dist.get_backend() == "gloo"

    prof = C.c_void_p()
    L.check(L.lib().tvr_profile_create(max(args.steps, 1), C.byref(prof)), "tvr_profile_create")

    chunk_fs = None
    if args.chunk > 0 and args.chunk_stream:
        from jittor_myc_nerfs_amd import FrameStream
        chunk_fs = FrameStream(model, white_bg=True, N_samples=S, eps_T=args.eps_T)
        chunk_out = (torch.empty((n_mine, 3), device=device), torch.empty((n_mine,), device=device))

    def step(s, profile=None, stats=None):
        rays = step_rays[s % n_patterns]
        if args.chunk > 0 and args.chunk_stream:                    # the same chunk loop with two calls in flight (render.FrameStream: call k on stream k % 2)
            rgb, depth = chunk_out
            for c0 in range(0, rays.shape[0], args.chunk):          # every call renders into its slice of one frame-sized output pair
                chunk_fs.submit(rays[c0:c0 + args.chunk], out=(rgb[c0:c0 + args.chunk], depth[c0:c0 + args.chunk]))
            chunk_fs.flush()
        elif args.chunk > 0:                                        # renderer.py:16-25 chunk loop, one tvr_render per chunk, no host sync
            outs = [model.render_rays(rays[c0:c0 + args.chunk], white_bg=True, N_samples=S, eps_T=args.eps_T, stats=stats)
                    for c0 in range(0, rays.shape[0], args.chunk)]
            rgb, depth = torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])
        else:
            rgb, depth = model.render_rays(rays, white_bg=True, N_samples=S, eps_T=args.eps_T, stats=stats, profile=profile, out=send_views)
        if dist_on:
            if args.chunk > 0:
                send_views[0].copy_(rgb)
                send_views[1].copy_(depth)
            if gloo:                                                # rehearsal path only
                parts = [torch.empty_like(mine) for _ in range(world)]
                dist.all_gather(parts, mine)
                return shard_unpermute(torch.cat(parts), R_step, world, cap, TILE)
            dist.all_gather_into_tensor(gathered, mine)
            return shard_unpermute(gathered, R_step, world, cap, TILE)      # undo the tile interleave: two strided copies (rgb, depth)
        return rgb, depth

    # A rank's share of a split frame is 2.6 ms at N = 8: the step then runs through ShardedFramePipeline (render.py) — the exchange (all_gather + un-permute) of frame k
    # on a side stream behind frame k + 1's kernels (two send / receive buffer pairs); optionally (--graph) the four launches of the render replayed as ONE hipGraph
    # per pose.  Emulation (--emulate-world, one process): the exchange is its device-side half at N-way sizes ("local").
    use_pipe = split > 1 or args.one_rank_exchange
    use_pipe = use_pipe and args.chunk == 0 and not args.no_pipeline
    pipe = None
    if use_pipe:
        pipe = ShardedFramePipeline(model, R_step, rank if dist_on else 0, world if dist_on else split, tile=TILE, white_bg=True, N_samples=S, eps_T=args.eps_T,
                                    exchange=("dist" if dist_on else (None if args.no_overlap_exchange else "local")), graph=args.graph, two_in_flight=args.two_shares_in_flight)

        def pipe_step(s):
            return pipe.submit(s % n_patterns, step_rays[s % n_patterns])

    # (20.64 ms per step against 20.62 ms of kernels).  When a rank's share is a fraction of a frame (strong split / --emulate-world: 2.7 ms
    # of kernels at N = 8) the timed region runs without them and the kernel durations are taken in K more steps afterwards.
    prof_in_region = split == 1
    prof_w = C.c_void_p()
    L.check(L.lib().tvr_profile_create(max(args.warmup, 1), C.byref(prof_w)), "tvr_profile_create")
    for s in range(args.warmup):
        step(s, profile=prof_w if prof_in_region else None)
    # Round 6, opt-in (--autotune): whether a frame in pieces beats one launch set depends on the card and on how long it has been under load: a library user
    # can decide it ON the card (model.autotune_render_pieces: 2 x 8 frames in each form, alternating).  NOT the default of this bench: the chip slows down under sustained load
    # (one launch set: 18.7 -> 19.25 ms per frame within the first second on one box; pieces: flat), so a decision taken in the first second can be the wrong one for the
    # stream that follows, and its 34 extra frames push the timed region itself into the slower regime (19.50 - 19.55 ms against 19.29 - 19.40 for the same form without them).
    pieces_autotune = None
    if args.autotune and args.pieces is None and split == 1 and args.chunk == 0 and not dist_on and hasattr(model, "autotune_render_pieces"):
        pieces_autotune = model.autotune_render_pieces(step_rays, white_bg=True, N_samples=S, eps_T=args.eps_T)
        for s in range(min(args.warmup, 2)):                        # (and the chosen form is warm again)
            step(s)
    if pipe is not None:                                            # captures one graph per (pose, send buffer) outside the timed region
        for pat in range(n_patterns):
            pipe.prepare(pat, step_rays[pat])
        for s in range(max(args.warmup, 2)):
            pipe_step(s)
        pipe.flush()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    # the interpreter's cyclic collector is parked for the timed region: a full collection of this process's heap takes ~40 ms of host time
    # (TVR_BENCH_TRACE=1 shows it as one step call that long), invisible beside 20 ms steps, 1 ms per step beside a 2.7 ms share of a frame
    # occupancy counters and the in-kernel clock probes (TVR_STAT_*) of exactly the timed launches: a handful of atomics at the END of each kernel
    stats = torch.zeros(8, dtype=torch.int64, device=device)
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    trace = [] if os.environ.get("TVR_BENCH_TRACE") else None
    for s in range(args.steps):
        if pipe is not None:
            pipe_step(args.warmup + s)
        else:
            # per-kernel events on every FOURTH timed step: a frame in 21 pieces records 84 events, +0.05 ... 0.15 ms per frame (scripts/debug/instr_cost.py); kernel_ms is
            # the average over the recorded steps (kernel_ms.calls of them), the in-kernel statistics cover every step
            step(args.warmup + s, profile=prof if (prof_in_region and s % 4 == 0) else None, stats=stats)
        if trace is not None:
            trace.append(time.perf_counter() - t0)
    if pipe is not None:
        pipe.flush()
    torch.cuda.synchronize()
    if trace is not None:
        print("host time after each step call (ms):", [round(t * 1e3, 2) for t in trace], "after sync: %.2f" % ((time.perf_counter() - t0) * 1e3), file=sys.stderr)
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gc.enable()
    if not prof_in_region:
        keep_pieces = model.render_piece_rays
        model.render_piece_rays = 0                                 # kernel durations of launches that have the chip (a 320 000-ray share would go out in pieces: see below)
        for s in range(args.steps):
            step(args.warmup + s, profile=prof, stats=stats if pipe is not None else None)    # (the graph replays carry neither events nor counters)
        torch.cuda.synchronize()
        model.render_piece_rays = keep_pieces
        model._ensure_scene()
    if dist_on:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    ms = (C.c_float * 3)()
    n_calls = L.lib().tvr_profile_read(prof, C.byref(ms))
    L.check(n_calls, "tvr_profile_read")
    k_ms = [ms[i] / max(n_calls, 1) for i in range(3)]
    if n_calls == 0:                                               # chunked mode records no per-kernel events
        k_ms = [0.0, 0.0, 0.0]
    # Round 6: a call of this size is rendered in PIECES on two library-owned streams (include/tvr.h) — the kernels of two pieces overlap, so the event durations of the
    # timed launches (k_ms above: sums over a frame's pieces) are durations under co-scheduling, not of a kernel that has the chip.  The roofline of a KERNEL is taken
    # from K more frames behind the timed region rendered as ONE launch set each (pieces off: the round-5 path, same library), with their own counters and clock probes.
    pieces_now = L.lib().tvr_scene_get_render_pieces(model._ensure_scene())
    n_pieces = 1
    if pieces_now > 0 and n_mine >= 6 * pieces_now:                 # (tvr_api.hip TVR_MIN_PIECES)
        k_ = (n_mine + pieces_now // 2) // pieces_now
        pr_ = ((n_mine + k_ - 1) // k_ + 511) // 512 * 512
        n_pieces = (n_mine + pr_ - 1) // pr_
    k_ms_timed, stats_timed = list(k_ms), stats
    if not prof_in_region:
        n_pieces = 1                                                # (the kernel durations above were taken with pieces off)
    if n_pieces > 1 and args.chunk == 0 and split == 1:
        keep = model.render_piece_rays
        model.render_piece_rays = 0
        prof_s = C.c_void_p()
        L.check(L.lib().tvr_profile_create(max(args.steps, 1), C.byref(prof_s)), "tvr_profile_create")
        stats = torch.zeros(8, dtype=torch.int64, device=device)
        step(0)
        torch.cuda.synchronize()
        ts0 = time.perf_counter()
        for s in range(args.steps):
            step(args.warmup + s, profile=prof_s, stats=stats)
        torch.cuda.synchronize()
        one_piece_ms = (time.perf_counter() - ts0) / args.steps * 1e3
        n_s = L.lib().tvr_profile_read(prof_s, C.byref(ms))
        L.check(n_s, "tvr_profile_read")
        k_ms = [ms[i] / max(n_s, 1) for i in range(3)]
        L.lib().tvr_profile_destroy(prof_s)
        model.render_piece_rays = keep
        model._ensure_scene()
    else:
        one_piece_ms = None

    check = None
    if args.check and dist_on:                                     # outside the timed region
        ok = True
        for pat in range(n_patterns):
            rgb_g, depth_g = step(pat)
            rgb1, depth1 = model.render_rays(full_rays[pat], white_bg=True, N_samples=S, eps_T=args.eps_T)
            ok = ok and torch.equal(rgb_g, rgb1) and torch.equal(depth_g, depth1)
            if pipe is not None:                                    # ... and the pipelined form (graph replay, exchange on the side stream), one frame behind
                prev = pipe_step(pat)
                if pat > 0:
                    ok = ok and torch.equal(prev[0], last1[0]) and torch.equal(prev[1], last1[1])
                last1 = (rgb1, depth1)
        if pipe is not None:
            fin = pipe.flush()
            ok = ok and torch.equal(fin[0], last1[0]) and torch.equal(fin[1], last1[1])
        flag = torch.tensor([1 if ok else 0], device=device)
        if gloo:
            flag = flag.cpu()
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) != 1:
            raise SystemExit("bench.py --check: the gathered frame differs from the single-rank render")
        check = "gathered == single-rank render, bit for bit"

    # BASELINE configs[2]'s own yardstick: the SAME frame rendered by one rank alone (outside the timed region; every rank does it on its own
    # GPU so that nobody idles at the next barrier, rank 0's time is reported).  SURVEY 8e: efficiency = t_1 / (N t_N).
    strong_split = None
    if strong:
        torch.cuda.synchronize()
        dist.barrier()
        t1s = []
        for s in range(max(3, min(args.steps, 10)) + 1):
            torch.cuda.synchronize()
            ta = time.perf_counter()
            model.render_rays(full_rays[s % n_patterns], white_bg=True, N_samples=S, eps_T=args.eps_T)
            torch.cuda.synchronize()
            t1s.append(time.perf_counter() - ta)
        t1 = float(np.median(t1s[1:]))
        strong_split = {"t1_ms": t1 * 1e3, "tN_ms": dt / args.steps * 1e3, "N": world, "t1_over_N_tN": t1 / (world * dt / args.steps),
                        "note": "t1 = the whole frame on rank 0's GPU alone, median of %d synchronised calls behind the timed region; tN = ms_per_step "
                                "(max over ranks, exchange and un-permute included)" % (len(t1s) - 1)}
        dist.barrier()

    st = stats.cpu().numpy().astype(np.float64) / max(args.steps, 1)      # per step (this rank): counters of the timed launches themselves
    m_eval, m_bbox, m_app = float(st[0]), float(st[1]), float(st[2])

    def probe_clock(i_clk, i_ref):                                  # s_memtime (shader clock) over s_memrealtime (100 MHz), summed over workgroups
        return 0.1 * st[i_clk] / st[i_ref] if st[i_ref] > 0 else None
    ck_march_probe, ck_shade_probe = probe_clock(4, 5), probe_clock(6, 7)

    # roofline.traffic: rocprofv3 --pmc passes over a 2-step run of this same bench, in CHILD processes (spawned, never exec'd from this
    # process), after the timed region — run before it, the profiler sessions left every later launch of this process ~0.5 ms slower
    # (24.0 vs 22.5 ms per step at identical kernel times), which would have distorted `value`.
    if args.pmc == "auto" and world == 1 and rank == 0 and default_workload:
        extra = (["--alpha-mask", str(args.alpha_mask)] if args.alpha_mask else []) + (["--eps-T", str(args.eps_T)] if args.eps_T is not None else [])
        pmc, pmc_source = collect_pmc(extra + ["--pieces", "0"])     # per-dispatch counters of the one-launch-set kernels (the roofline's kernels)

    rays_job = R_step if args.emulate_world <= 1 else n_mine       # emulation reports the share that was rendered
    value = rays_job * S * args.steps / dt
    has_mask = model.alphaMask is not None
    t_march, t_shade = k_ms[0] * 1e-3, k_ms[1] * 1e-3
    FLOP_APP = 8.0e4            # algorithmic FLOP per appearance sample (basis 7 776 + MLP 71 936 + PE)
    FLOP_APP_EXEC = n_prod * 2 * (32 * 144 + 128 * 160 + 128 * 128)        # executed on the matrix cores: 3 fp16 products (--arith f32), padded tiles (layer 3 runs as fp32 VALU FMAs)
    if args.model == "REFTensoRF":                                         # + four 144 -> {3,3,1,1} heads, 151-input layer 1
        FLOP_APP += 2 * 8 * 144 + 2 * 128
        FLOP_APP_EXEC += 3 * 2 * 32 * 144
    # the render path's shade kernel: tvr_shade16.hip's shade16_kernel (16x16x32 tiles) for TensorVMSplit in the default arithmetic, else tvr_shade.hip's
    on16 = args.model == "TensorVMSplit" and args.arith == "f32"
    pm, ps = _pmc_kernel(pmc, "march_kernel<false"), (_pmc_kernel(pmc, "shade16_kernel") if on16 else _pmc_kernel(pmc, "shade_kernel<0, 0"))

    def hbm(c):
        return (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 if ("FETCH_SIZE" in c and "WRITE_SIZE" in c) else None

    def clock_ghz(c, t):                                           # GRBM_GUI_ACTIVE sums the 8 XCDs (MI355X_MICROARCH.md 'DVFS give-back')
        return c["GRBM_GUI_ACTIVE"] / 8.0 / t / 1e9 if ("GRBM_GUI_ACTIVE" in c and t > 0) else None

    # march: a cache-bandwidth gather.  Per evaluated density sample 768 B of plane texels come through the vector L1 (TA path) and 384 B of
    # line texels from LDS; per ray 24 B in + 16 B out.  The roof that binds is the L1 -> register path, 64 B/clk/CU (CDNA3/4 vector L1 data
    # return width; the guide gives no L1 figure, its L2 aggregate is 34.5 TB/s) x 256 CUs x the clock measured under this load.
    l1_bytes = 40.0 * n_mine + 768.0 * m_eval + (32.0 * m_bbox if has_mask else 0.0)
    lds_bytes = 384.0 * m_eval
    # The clock of the roof is the one the kernel ran at IN THE TIMED LAUNCHES (in-kernel s_memtime / s_memrealtime probe); the PMC child pass's
    # GRBM_GUI_ACTIVE / 8 / (that pass's own kernel duration) is kept beside it as a cross-check.  peak = 256 CUs x 64 B/clk (the vector L1's
    # data-return width) x that clock: bytes <= 64 B x CU-cycles, so frac <= 1 by construction.
    ck_m = ck_march_probe or clock_ghz(pm, pm.get("_dur_s", 0.0)) or 2.1
    L1_NOMINAL = 64.0
    L1_PROBE = 48.9              # B/clk/CU a micro-probe of wave-level dwordx4 loads that all hit L1 sustains with 16 waves per CU (scripts/hwprobe/ta_rate.hip,
                                 # profiles/r02_l1_rate_probe.txt: 21.0 cycles per 1-KB load).  A measurement of ONE access shape, not a roof: the march kernel's
                                 # own shape (16 contiguous 64-B segments per load, 16 waves) runs slightly above it
    l1_peak = 256 * L1_NOMINAL * ck_m                               # GB/s
    roof_march = {"kernel": "march_kernel<false>", "bound": "l1", "achieved": l1_bytes / t_march / 1e9 if t_march > 0 else None,
                  "peak": l1_peak, "unit": "GB/s", "frac": l1_bytes / t_march / 1e9 / l1_peak if t_march > 0 else None,
                  "traffic": hbm(pm), "algorithmic_bytes_per_launch": l1_bytes + lds_bytes, "l1_bytes_per_launch": l1_bytes,
                  "lds_bytes_per_launch": lds_bytes, "lds_GBps": lds_bytes / t_march / 1e9 if t_march > 0 else None,
                  "ratio_to_l1_microprobe_48.9B_per_clk": l1_bytes / t_march / 1e9 / (256 * L1_PROBE * ck_m) if t_march > 0 else None,
                  "clock_GHz": ck_m, "clock_source": ("in-kernel probe over the timed launches: 0.1 GHz x sum(s_memtime) / sum(s_memrealtime)" if ck_march_probe else
                                                      ("GRBM_GUI_ACTIVE / 8 / kernel duration, both of the PMC child pass" if clock_ghz(pm, pm.get("_dur_s", 0.0)) else "nominal 2.1 (no probe, no PMC pass)")),
                  "clock_GHz_pmc_pass": clock_ghz(pm, pm.get("_dur_s", 0.0)),
                  "vs_hbm_8TBps": (l1_bytes + lds_bytes) / t_march / 1e9 / 8000.0 if t_march > 0 else None, "ms": k_ms[0],
                  "lds_bank_conflict_frac": (pm["SQ_LDS_BANK_CONFLICT"] / pm["SQ_LDS_IDX_ACTIVE"]) if pm.get("SQ_LDS_IDX_ACTIVE") else None,
                  "note": "40 B/ray + 768 B through L1 + 384 B through LDS per density sample actually evaluated (+32 B per alpha-mask lookup); peak = 256 CUs x "
                          "64 B/clk (vector-L1 data-return width) x the clock measured inside the timed launches.  The 17 MB of density factors are L2 / Infinity-Cache "
                          "resident, so HBM is not the roof (vs_hbm_8TBps > 1 by construction, kept for SURVEY 8d's formula)"}
    # wave-level dwordx4 loads (1 KB each) per 32-entry tile of the shade kernel: counted by the PMC pass when there is one, else the source's own count
    loads_per_tile = (ps["SQ_INSTS_VMEM_RD"] / (m_app / 32.0)) if (ps.get("SQ_INSTS_VMEM_RD") and m_app > 0) else float(SHADE_LOADS_PER_TILE)
    SHADE_L1_B = loads_per_tile * 1024 / 32.0
    ck_s = ck_shade_probe or clock_ghz(ps, ps.get("_dur_s", 0.0)) or 1.7
    ach_shade = FLOP_APP * m_app / t_shade / 1e12 if t_shade > 0 else None
    if on16:                    # executed on the 16x16x32 tiles: basis 32 rows x 160 k, layer 1 128 x 160, layer 2 128 x 128, three products each
        FLOP_APP_EXEC = 3 * 2 * (32 * 160 + 128 * 160 + 128 * 128)
    roof_shade = {"kernel": "shade16_kernel<false>" if on16 else "shade_kernel<0,0,%s>" % ("true" if args.model == "REFTensoRF" else "false"), "bound": "mfma", "achieved": ach_shade,
                  "peak": 2500.0, "unit": "TFLOP/s", "frac": ach_shade / 2500.0 if ach_shade else None,
                  "frac_vs_fp32class_ceiling": ach_shade / (2500.0 / 3.0) if ach_shade else None,
                  # what the matrix pipe sustains on toggling operands (power envelope): scripts/hwprobe/mfma_clock.hip, profiles/r02_mfma_clock_probe.txt
                  "sustained_mfma_TFLOPs": 1704.0, "frac_vs_sustained_mfma": ach_shade / 1704.0 if ach_shade else None,
                  "executed_frac_vs_sustained_mfma": FLOP_APP_EXEC * m_app / t_shade / 1e12 / 1704.0 if t_shade > 0 else None,
                  "traffic": hbm(ps), "algorithmic_flops_per_launch": FLOP_APP * m_app, "ms": k_ms[1],
                  "executed_mfma_TFLOPs": FLOP_APP_EXEC * m_app / t_shade / 1e12 if t_shade > 0 else None,
                  "gather_algorithmic_GBps": 3456.0 * m_app / t_shade / 1e9 if t_shade > 0 else None,
                  "mfma_busy_frac": (ps["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (ps["GRBM_GUI_ACTIVE"] / 8.0)) if ps.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in ps else None,
                  "mfma_valu_coexec_frac_of_busy": (ps["SQ_VALU_MFMA_COEXEC_CYCLES"] / ps["SQ_VALU_MFMA_BUSY_CYCLES"]) if ps.get("SQ_VALU_MFMA_BUSY_CYCLES") else None,
                  "valu_insts_per_32_entry_tile": (ps["SQ_INSTS_VALU"] / (m_app / 32.0)) if ps.get("SQ_INSTS_VALU") and m_app > 0 else None,
                  "clock_GHz": ck_s, "clock_source": "in-kernel probe over the timed launches" if ck_shade_probe else "PMC child pass or nominal",
                  "clock_GHz_pmc_pass": clock_ghz(ps, ps.get("_dur_s", 0.0)), "vmem_loads_per_32_entry_tile": loads_per_tile,
                  "valu_busy_frac": (ps["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / (ps["GRBM_GUI_ACTIVE"] / 8.0)) if ps.get("GRBM_GUI_ACTIVE") and "SQ_ACTIVE_INST_VALU" in ps else None,
                  "lds_busy_frac": (ps["SQ_LDS_IDX_ACTIVE"] / 256.0 / (ps["GRBM_GUI_ACTIVE"] / 8.0)) if ps.get("GRBM_GUI_ACTIVE") and "SQ_LDS_IDX_ACTIVE" in ps else None,
                  "lds_bank_conflict_frac": (ps["SQ_LDS_BANK_CONFLICT"] / ps["SQ_LDS_IDX_ACTIVE"]) if ps.get("SQ_LDS_IDX_ACTIVE") else None,
                  "pmc_counters": {k: ps[k] for k in sorted(ps)},
                  # the second roof of this kernel: every appearance sample pulls 3456 B of taps + 576 B of basis fragments (18 KB per 32-entry tile)
                  # + 36 B of queue entry / view direction through the vector L1
                  # 123 wave-level dwordx4 loads (1 KB each) per 32-entry tile: 108 taps, 9 basis fragments (lo parts), 6 entry / direction
                  "l1": {"bytes_per_launch": SHADE_L1_B * m_app, "achieved_GBps": SHADE_L1_B * m_app / t_shade / 1e9 if t_shade > 0 else None,
                         "peak_GBps": 256 * L1_NOMINAL * ck_s,
                         "frac": SHADE_L1_B * m_app / t_shade / 1e9 / (256 * L1_NOMINAL * ck_s) if t_shade > 0 else None,
                         "ratio_to_l1_microprobe_48.9B_per_clk": SHADE_L1_B * m_app / t_shade / 1e9 / (256 * L1_PROBE * ck_s) if t_shade > 0 else None},
                  "note": f"{FLOP_APP / 1e3:.1f} kFLOP per appearance sample (algorithmic, fp32 semantics) against the dense f16 MFMA peak (2.5 PFLOP/s); "
                          "fp32-class arithmetic on this chip needs 3 fp16 products per fp32 product (hi/lo split; the fp32-input MFMA runs at 1/16 rate), "
                          f"so the ceiling for this arithmetic is peak / 3 = 833 TFLOP/s (frac_vs_fp32class_ceiling); executed on padded tiles: {FLOP_APP_EXEC} FLOP "
                          "per sample; the gather moves 3456 B per sample through L1"}
    for r_, c_ in ((roof_march, pm), (roof_shade, ps)):
        r_["traffic_source"] = pmc_source
        if c_.get("SQ_WAVE_CYCLES"):                                  # where a wave's cycles go (quad-cycle units; the three are disjoint)
            r_["wave_cycles_frac"] = {k: c_[n] / c_["SQ_WAVE_CYCLES"] for k, n in (("waiting_on_waitcnt_or_barrier", "SQ_WAIT_ANY"), ("issue_stalled", "SQ_WAIT_INST_ANY"),
                                                                            ("issuing", "SQ_ACTIVE_INST_ANY")) if n in c_}
    if n_pieces > 1 and one_piece_ms is not None:
        for r_, i_ in ((roof_march, 0), (roof_shade, 1)):
            r_["measured"] = (f"{args.steps} frames behind the timed region, each rendered as ONE launch set (--pieces 0: the kernel has the chip), HIP events + in-kernel "
                              f"counters and clock probes of those launches: {one_piece_ms:.3f} ms per frame that way.  In the timed region the frame goes out as {n_pieces} "
                              f"pieces on two streams ({dt / args.steps * 1e3:.3f} ms per frame) and a kernel's launches overlap the other stream's: in_timed_region")
            t_ = k_ms_timed[i_] * 1e-3
            r_["in_timed_region"] = {"ms_sum_over_pieces": k_ms_timed[i_], "pieces": n_pieces,
                                     "achieved": (r_["achieved"] * (k_ms[i_] / k_ms_timed[i_])) if (r_.get("achieved") and k_ms_timed[i_] > 0) else None,
                                     "frac": (r_["frac"] * (k_ms[i_] / k_ms_timed[i_])) if (r_.get("frac") and k_ms_timed[i_] > 0) else None,
                                     "note": "the same algorithmic work over the summed durations of the timed launches, which share the chip with the other stream's kernels"}
    dominant = roof_shade if k_ms[1] >= k_ms[0] else roof_march
    mode = ("N>1 weak: N frames per step" if (world > 1 and not strong) else "ONE frame per step at every N (N>1: split over the ranks, BASELINE configs[2])")
    result = {
        "metric": "ray_samples_per_sec", "value": value, "unit": "ray-samples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak" if (world > 1 and not strong) else "strong",          # default: ONE frame per step at every N (total work fixed)
        "vs_baseline": None, "dtype": {"f32": "f32 (emulated: 3 x f16-split MFMA products, fp32 accumulate; fp32 VALU elsewhere)",
                                       "f16act": "f16 activations x f32-class weights (2 x f16 MFMA products, fp32 accumulate; fp32 VALU elsewhere) - opt-in mode, not the headline",
                                       "f16": "f16 operands (1 MFMA product, fp32 accumulate; fp32 VALU elsewhere) - opt-in mode, not the headline"}[args.arith], "data": "synthetic",
        "config": {"workload": ("TensorVMSplit 300^3 (16/48 comps, MLP_Fea 150-128-128-3)" if args.model == "TensorVMSplit" else
                                "REFTensoRF 300^3 (16/48 comps, 4 heads on h, MLP_Fea_Ref 151-128-128-3)") +
                               f", {args.img}x{args.img} rays x 512 samples/ray (BASELINE configs[1]); {mode}, {TILE}-ray tiles round-robin, one RCCL "
                               "all_gather of [4 cap] fp32 (rgb block + depth block, written in place by the render) + two strided un-permute copies",
                   "scene": "synthetic scene A (SURVEY 8d), no alpha mask, white_bg",
                   "rays_per_step": rays_job, "samples_per_ray": S, "rays_per_rank": n_mine,
                   "eps_T": float(model.rayMarch_weight_thres) if args.eps_T is None else args.eps_T, "tile": TILE,
                   "chunk": args.chunk if args.chunk > 0 else n_mine, "alpha_mask": args.alpha_mask, "emulate_world": args.emulate_world, "arith": args.arith},
        "rays_per_sec": rays_job * args.steps / dt,
        "effective": {"density_samples_evaluated_per_sec": m_eval * world * args.steps / dt,
                      "appearance_samples_per_sec": m_app * world * args.steps / dt,
                      "frac_samples_evaluated": m_eval / (n_mine * S),
                      # (round 6: this was called frac_samples_in_box; the counter only runs over the chunks a ray visits before it terminates, so it is NOT the in-box
                      #  fraction of all nominal samples — without an alpha mask it equals frac_samples_evaluated by construction)
                      "frac_samples_in_box_of_visited_chunks": m_bbox / (n_mine * S),
                      "app_samples_per_ray": m_app / n_mine},
        "kernel_ms": {"march": k_ms_timed[0], "shade": k_ms_timed[1], "composite": k_ms_timed[2], "calls": n_calls, "pieces_per_call": n_pieces,
                      "note": ("sums over the pieces of a call of the HIP-event durations of the TIMED launches; two pieces are in flight at a time on two streams, so these are "
                               "durations under co-scheduling (march + shade + composite > ms_per_step)" if n_pieces > 1 else "HIP events around each kernel of the timed launches")
                              + ("; recorded on every fourth timed step (`calls` of them)" if prof_in_region else "")},
        "roofline": dominant,
        "roofline_all": {"march": roof_march, "shade": roof_shade},
    }
    if pieces_autotune is not None:
        result["config"]["pieces_autotune"] = dict(pieces_autotune, note="--autotune: decided on this card in the warm-up (model.autotune_render_pieces)")
    if pipe is not None:
        result["split_step"] = {"pipeline": "ShardedFramePipeline", "hipgraph": pipe.use_graph, "exchange": pipe.exchange, "two_shares_in_flight": bool(pipe.two),
                                "note": "the exchange of frame k (all_gather + un-permute; 'local' = its device-side half at N-way sizes, one process) runs on a side stream "
                                        "behind frame k + 1's kernels; hipgraph: the per-rank render replayed as one graph per pose"}
    if dist_on:
        result["exchange_backend"] = dist.get_backend() + (" (one-rank rehearsal: the process group has ONE member)" if world == 1 else "")
    if check is not None:
        result["check"] = check
    if strong_split is not None:
        result["strong_split"] = strong_split
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(arrs, A, fr[0], with_c=(args.model == "TensorVMSplit"))
    elif rank == 0:
        result["cpu_baseline"] = None
    # The other single-GPU lines of BASELINE.json, measured by THIS run (child processes of this script, after the timed region, so that whoever runs the
    # default command also gets them): configs[3] = the same frame as 157 direct 4096-ray tvr_render calls (train.py's batch size, no chunk merging),
    # configs[4] = the JNeRF Instant-NGP alt path.  Informational: `value` above is configs[1].
    if rank == 0 and world == 1 and default_workload and not args.no_extras and args.arith == "f32":
        result["arith_modes"] = arith_modes(model, step_rays, S, args.eps_T, m_app)
        result["frame_stream_two_in_flight"] = frame_stream(model, step_rays, S, args.eps_T)
    if rank == 0 and world == 1 and default_workload and not args.no_extras and not args.no_cpu_baseline:
        result["other_configs"] = other_configs()
    L.lib().tvr_profile_destroy(prof)
    L.lib().tvr_profile_destroy(prof_w)
    if rank == 0:
        emit(result)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
