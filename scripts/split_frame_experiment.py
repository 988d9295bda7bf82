#!/usr/bin/env python3
"""Round 6, review item 2: the overlap FrameStream measured ACROSS frames (18.8 vs 19.7 ms), brought INSIDE one frame.

One 640 000-ray frame rendered as K pieces of consecutive rays, piece k on stream k % n_streams with that stream's own scratch, into slices of ONE output; the
streams fork from and join the caller's stream by events.  Per-ray results do not depend on the batch a ray arrives in, so the frame must equal the plain
render bit for bit.  Interleaved over the configurations (same process, same box, round-robin), 8 poses per measurement.

    python scripts/split_frame_experiment.py [--rounds 3] [--frames 16]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--model", default="TensorVMSplit")
    ap.add_argument("--K2", default="2,3,4,6,8,12,16,24", help="piece counts tried on two streams")
    ap.add_argument("--K3", default="6,9,12", help="piece counts tried on three streams")
    ap.add_argument("--stagger", default="5:0.1,9:0.06", help="K:first-piece-fraction pairs (two streams)")
    ap.add_argument("--pieces", default="", help="piece sizes in rays (uniform pieces, the last one shorter), two streams")
    ap.add_argument("--lists", default="", help="explicit piece-size lists, ';'-separated, each 'name=a,b,c*N,...' (c*N repeats; the remainder of the frame is cut into pieces of the last size)")
    ap.add_argument("--cpieces", default="", help="piece sizes handed to the LIBRARY (tvr_scene_set_render_pieces: the shipped mechanism), one configuration each")
    ap.add_argument("--rays", type=int, default=0, help="render only the first N rays of each frame (a rank's share)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    model, arrs, A = bench.build_model(dev, args.model)
    S = A["N_samples"]
    fr = [f.to(dev) for f in bench.frames(A)]
    if args.rays:
        fr = [f[:args.rays].contiguous() for f in fr]
    n = fr[0].shape[0]
    streams = [torch.cuda.Stream(dev) for _ in range(3)]
    rgb = torch.empty((n, 3), device=dev)
    depth = torch.empty((n,), device=dev)

    def plain(rays, pieces=0):
        model.render_piece_rays = pieces                  # 0: ONE launch set per call (the round-5 path); every Python-level split below runs its pieces that way too
        return model.render_rays(rays, white_bg=True, N_samples=S, out=(rgb, depth))

    def split(rays, K, n_streams=2, first_frac=None, align=4096, piece=0, sizes=None):
        cur = torch.cuda.current_stream(dev)
        # piece boundaries: K pieces of (roughly) equal size, multiples of `align` rays; optionally a smaller first piece (a stagger)
        if sizes:
            edges, a = [0], 0
            for sz in sizes:
                a = min(n, a + sz)
                edges.append(a)
            while a < n:
                a = min(n, a + sizes[-1])
                edges.append(a)
        elif piece:
            edges = list(range(0, n, piece)) + [n]
        elif first_frac is None:
            edges = [min(n, ((n * k // K + align - 1) // align) * align) for k in range(K)] + [n]
        else:
            first = int(n * first_frac / align) * align
            rest = n - first
            edges = [0] + [min(n, first + ((rest * k // (K - 1) + align - 1) // align) * align) for k in range(K - 1)] + [n]
            edges = sorted(set(edges))
        fork = torch.cuda.Event()
        fork.record(cur)
        for s in streams[:n_streams]:
            s.wait_event(fork)
        for k in range(len(edges) - 1):
            a, b = edges[k], edges[k + 1]
            if b <= a:
                continue
            s = k % n_streams
            model.render_piece_rays = 0
            with torch.cuda.stream(streams[s]):
                model.render_rays(rays[a:b], white_bg=True, N_samples=S, out=(rgb[a:b], depth[a:b]), scratch_slot=s)
        for s in streams[:n_streams]:
            cur.wait_stream(s)
        return rgb, depth

    configs = [("plain", lambda r: plain(r))]
    for cp in [int(x) for x in args.cpieces.split(",") if x]:
        configs.append((f"library pieces of {cp} rays", lambda r, cp=cp: plain(r, cp)))
    for K in [int(x) for x in args.K2.split(",") if x]:
        configs.append((f"K={K} 2 streams", lambda r, K=K: split(r, K, 2, align=512 if K > 100 else 4096)))
    for K in [int(x) for x in args.K3.split(",") if x]:
        configs.append((f"K={K} 3 streams", lambda r, K=K: split(r, K, 3, align=512 if K > 100 else 4096)))
    for kv in [x for x in args.stagger.split(",") if x]:
        K, ff = int(kv.split(":")[0]), float(kv.split(":")[1])
        configs.append((f"K={K} 2 streams, first piece {ff:g}", lambda r, K=K, ff=ff: split(r, K, 2, first_frac=ff)))

    for ps in [int(x) for x in args.pieces.split(",") if x]:
        configs.append((f"pieces of {ps} rays, 2 streams", lambda r, ps=ps: split(r, 0, 2, piece=ps)))

    for spec in [x for x in args.lists.split(";") if x]:
        name, body = spec.split("=")
        sizes = []
        for tok in body.split(","):
            if "*" in tok:
                v, r = tok.split("*")
                sizes += [int(v)] * int(r)
            else:
                sizes.append(int(tok))
        configs.append((f"list {name}", lambda r, sizes=sizes: split(r, 0, 2, sizes=sizes)))

    # correctness first: every configuration equals the plain render bit for bit (pose 3)
    ref = [t.clone() for t in plain(fr[3])]
    torch.cuda.synchronize()
    for name, fn in configs[1:]:
        rgb.fill_(-1)
        depth.fill_(-1)
        fn(fr[3])
        torch.cuda.synchronize()
        assert torch.equal(rgb, ref[0]) and torch.equal(depth, ref[1]), name
    print("every split configuration == plain render, bit for bit", flush=True)

    res = {name: [] for name, _ in configs}
    for rnd in range(args.rounds):
        for name, fn in configs:
            for w in range(2):
                fn(fr[w])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for f in range(args.frames):
                fn(fr[f % len(fr)])
            torch.cuda.synchronize()
            res[name].append((time.perf_counter() - t0) / args.frames * 1e3)
        print(f"round {rnd}: " + "  ".join(f"{k}: {v[-1]:.2f}" for k, v in res.items()), flush=True)
    base = min(res["plain"])
    print(f"\n{'configuration':<40} {'ms per frame (min / median over rounds)':<42} vs plain")
    for name, v in res.items():
        v = sorted(v)
        print(f"{name:<40} {v[0]:7.3f} / {v[len(v) // 2]:7.3f} {'':<24} {100 * (v[0] / base - 1):+.2f} %")


if __name__ == "__main__":
    main()
