#!/usr/bin/env python3
"""GPU box: does tests/test_gpu_parity.py::test_frame_stream_serialises_around_a_scene_update have teeth?  The same schedule with FrameStream's drain disabled
(scene_settled forced True): frames whose re-pack ran beside the frame in flight on the other stream come out wrong."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from jittor_myc_nerfs_amd import FrameStream, synthetic
from test_gpu_parity import make_model

if os.environ.get("FS_BENCH_SCENE"):                           # the 800 x 800 bench frame: 20 ms per frame, the re-pack of frame k lands inside frame k - 1
    import bench
    m, arrs, A = bench.build_model(torch.device("cuda"))
    rays, S = bench.frames(A)[0].cuda(), 512
else:
    B = synthetic.SCENE_B
    arrs = synthetic.make_scene_arrays(B["gridSize"], B["aabb"])
    m = make_model(arrs, dict(synthetic.HYPER, near_far=B["near_far"], step_ratio=B["step_ratio"]))
    g = np.load(os.path.join(ROOT, "tests", "golden", "config1.npz"))
    rays, S = torch.cat([torch.tensor(g["rays"], device="cuda")] * 8), B["N_samples"]
w0 = m.app_plane[0].detach().clone()


def schedule(render):
    out = []
    for k in range(6):
        if k in (2, 3, 5):
            with torch.no_grad():
                m.app_plane[0].mul_(1.0 + 0.05 * k)
        out.append(render(k))
    return out


want = schedule(lambda k: m.render_rays(rays, white_bg=True, N_samples=S)[0].clone())
print("serial schedule, max |frame k - frame k-1|:", [float((want[k] - want[k - 1]).abs().max()) for k in range(1, 6)])
for drain in (True, False):
    with torch.no_grad():
        m.app_plane[0].copy_(w0)
    if not drain:
        m.scene_settled = lambda: True
    fs, got = FrameStream(m, white_bg=True, N_samples=S), []
    def sub(k):
        o = fs.submit(rays)
        if o is not None:
            got.append(o[0].clone())
    schedule(sub)
    got.append(fs.flush()[0].clone())
    print("drain" if drain else "NO drain", [float((a - b).abs().max()) for a, b in zip(got, want)])
