#!/usr/bin/env python3
"""GPU box (ONE card): what BASELINE configs[2] — the 800x800 frame of scene A split over N ranks — costs a rank, as far as one GPU can say.

    python3 scripts/strong_emulation.py [--steps 20] > profiles/r04_strong_emulation.json

  * kernels: `bench.py --emulate-world N` renders rank 0's share of the N-way split (tiles 0, N, 2N, ... of bench.py's shard tile) — N = 1, 2, 4, 8, each in a
    fresh child process; efficiency_kernels = t_1 / (N t_N) over the whole step (march + shade + composite + launch gaps).
  * exchange, device side: the two strided copies that undo the tile interleave of the gathered [N, 4 cap] buffer (the render writes its pixels
    straight into the send buffer: no pad copy), timed with HIP events at each N's own sizes, beside round 2's single index gather.
  * exchange, transport: a 2-rank `gloo` run of the real step (`bench.py --gpus 2 --scaling strong --check`, both ranks on this card).  gloo stages
    device tensors through the host, so its all_gather is an UPPER bound for RCCL over xGMI (10.2 MB per frame); the number is recorded, not used.
No multi-GPU node has been available to the build; the hardware curve stays unmeasured until the driver's SCALE run."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run_json(cmd, env=None, timeout=900):
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    if r.returncode != 0:
        raise SystemExit("failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout[-1500:], r.stderr[-3000:]))
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def device_side_exchange(worlds, R=640000, tile=None, reps=50):
    import torch
    sys.path.insert(0, ROOT)
    from jittor_myc_nerfs_amd import shard_capacity, shard_gather_index, shard_indices, shard_unpermute
    from jittor_myc_nerfs_amd.render import SHARD_TILE
    tile = tile or SHARD_TILE
    dev = torch.device("cuda", 0)
    out = {}
    for w in worlds:
        cap = shard_capacity(R, w, tile)
        n_mine = shard_indices(R, 0, w, tile).numel()
        gathered = torch.rand((w * 4 * cap,), device=dev)
        inv = shard_gather_index(R, w, tile, dev)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        t_new = t_old = 0.0
        for i in range(reps + 5):
            ev[0].record()
            rgb, depth = shard_unpermute(gathered, R, w, cap, tile)
            ev[1].record()
            img = gathered.view(-1, 4)[:w * cap].index_select(0, inv)        # round 2's form: one index gather of [N cap,4] (same bytes)
            ev[2].record()
            torch.cuda.synchronize()
            if i >= 5:
                t_new += ev[0].elapsed_time(ev[1])
                t_old += ev[1].elapsed_time(ev[2])
        out[str(w)] = {"cap_rays": cap, "rays_rank0": n_mine, "send_bytes": cap * 16, "gathered_bytes": w * cap * 16,
                       "pad_copy_ms": 0.0, "unpermute_ms": t_new / reps, "round2_index_gather_ms": t_old / reps,
                       "note": "no pad copy: the render kernels write rgb / depth straight into the send buffer; the un-permute is two strided copies"}
        del img, rgb, depth
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--skip-gloo", action="store_true")
    args = ap.parse_args()
    worlds = [1, 2, 4, 8]
    rows = {}
    for w in worlds:
        base = [sys.executable, BENCH, "--steps", str(args.steps), "--warmup", "3", "--no-cpu-baseline", "--pmc", "off"]
        d = run_json(base + (["--emulate-world", str(w), "--no-overlap-exchange"] if w > 1 else []))
        rows[str(w)] = {"rays": d["config"]["rays_per_rank"], "tile": d["config"]["tile"], "ms_per_step": d["ms_per_step"], "kernel_ms": d["kernel_ms"],
                        "clock_GHz": {"march": d["roofline_all"]["march"]["clock_GHz"], "shade": d["roofline_all"]["shade"]["clock_GHz"]}}
        if w > 1:      # the same step with the device-side half of the exchange (copy into the receive buffer + the two un-permute copies, at N-way sizes) on the side stream
            rows[str(w)]["ms_per_step_with_overlapped_exchange"] = run_json(base + ["--emulate-world", str(w)])["ms_per_step"]
        print("emulate-world %d: %.3f ms per step, kernels %s" % (w, d["ms_per_step"], d["kernel_ms"]), file=sys.stderr)
    t1 = rows["1"]["ms_per_step"]
    k1 = {k: rows["1"]["kernel_ms"][k] for k in ("march", "shade", "composite")}
    for w in worlds:
        r = rows[str(w)]
        r["efficiency_step"] = t1 / (w * r["ms_per_step"])
        r["efficiency_per_kernel"] = {k: k1[k] / (w * r["kernel_ms"][k]) if r["kernel_ms"][k] > 0 else None for k in k1}
    ex = device_side_exchange(worlds[1:])
    for w in worlds[1:]:
        r, e = rows[str(w)], ex[str(w)]
        r["efficiency_step_with_device_side_exchange"] = t1 / (w * r["ms_per_step_with_overlapped_exchange"])       # round 4: overlapped on a side stream (ShardedFramePipeline)
        r["efficiency_step_with_serial_device_side_exchange"] = t1 / (w * (r["ms_per_step"] + e["pad_copy_ms"] + e["unpermute_ms"]))   # round 3's form: behind the render, same stream
    out = {"what": "BASELINE configs[2] on ONE MI355X: rank 0's share of the N-way strong split of the 800x800 frame (kernels, --emulate-world), the device-side "
                   "half of the exchange, and a 2-rank gloo rehearsal; t_1 / (N t_N) per SURVEY 8e.  The RCCL all_gather itself is not measurable on one card.",
           "command": "python3 scripts/strong_emulation.py --steps %d" % args.steps, "emulation": rows, "exchange_device_side": ex}
    if not args.skip_gloo:
        env = dict(os.environ, TVR_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        port = 36500 + os.getpid() % 2000
        t0 = time.time()
        d = run_json([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                      BENCH, "--gpus", "2", "--steps", str(args.steps), "--warmup", "3", "--check", "--no-cpu-baseline"], env=env, timeout=1500)
        out["gloo_world2_one_card"] = {"ms_per_step": d["ms_per_step"], "strong_split": d.get("strong_split"), "check": d.get("check"), "kernel_ms": d["kernel_ms"],
                                       "wall_s": time.time() - t0,
                                       "note": "both ranks share the card (their kernels serialise) and gloo stages the 2 x 5.1 MB through the host: an upper bound of "
                                               "the step, recorded for the exchange path's correctness (--check) and order of magnitude, not as a scaling number"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
