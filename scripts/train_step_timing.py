"""Times the training step (forward + backward + Adam) at the reference's settings: TensorVMSplit 300^3, batch 4096 rays,
nSamples = min(1e6, cal_n_samples(reso, 0.5)) = 1039 (train.py:143-144, utils.py:61-62), MSE + regularisers (train.py:228-251)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import torch.distributed as dist
from jittor_myc_nerfs_amd import GradBucket, OctreeRender_trilinear_fast, TVLoss, shard_batch
# data-parallel: python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 scripts/train_step_timing.py
world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
if world > 1:
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    dist.init_process_group(os.environ.get("TVR_BENCH_BACKEND", "nccl"))
MODEL = os.environ.get("TVR_MODEL", "TensorVMSplit")        # REFTensoRF: what configs/Scar.txt:28 trains (+ the normal penalty, train.py:253-257)
m, arrs, A = bench.build_model(torch.device("cuda"), MODEL)
m.fused_mlp_training = bool(int(os.environ.get("TVR_FUSED_MLP", "1")))      # 0: round-1 path (library GEMMs for the MLP forward / dX)
m.static_training = bool(int(os.environ.get("TVR_STATIC", "1")))           # 1: tvr_train_forward / _backward (no host read); 0: the eager chain of autograd Functions
GRAPH = bool(int(os.environ.get("TVR_GRAPH", "0")))                        # 1: the whole step (forward, backward, regularisers, Adam) captured once, replayed
with torch.no_grad():                                  # start from a perturbed copy so that gradients are non-trivial
    for p in m.parameters():
        p.mul_(0.9)
fr = bench.frames(A)
allrays = torch.cat(fr[:4]).cuda()
with torch.no_grad():
    teacher, _, _ = bench.build_model(torch.device("cuda"), MODEL)
    allrgbs = torch.cat([teacher.render_rays(allrays[i:i + 640000], N_samples=512)[0] for i in range(0, allrays.shape[0], 640000)])
    del teacher
nS = int(np.linalg.norm(A["gridSize"]) / 0.5)
opt = torch.optim.Adam(m.get_optparam_groups(0.02, 0.001), betas=(0.9, 0.99), fused=bool(int(os.environ.get("TVR_FUSED_ADAM", "1"))), capturable=GRAPH)
tv = TVLoss()
g = None if GRAPH else torch.Generator(device="cuda").manual_seed(0)       # (under capture: the default generator, whose state torch registers with the graph)
bucket = GradBucket(m) if (world > 1 or os.environ.get("TVR_BUCKET")) else None    # all gradients in one 70 MB buffer: one all-reduce per step
def step():
    idx = torch.randint(0, allrays.shape[0], (4096,), device="cuda", generator=g)[shard_batch(4096, rank, world)]
    bucket.zero() if bucket else opt.zero_grad()
    rgb_map, _, _, _, _ = OctreeRender_trilinear_fast(allrays[idx], m, chunk=4096, N_samples=nS, white_bg=True, is_train=True)
    loss = torch.mean((rgb_map - allrgbs[idx]) ** 2)
    total = loss + 1e-4 * m.vector_comp_diffs() + 8e-5 * m.density_L1() + 0.1 * m.TV_loss_density(tv) + 0.01 * m.TV_loss_app(tv)
    if MODEL == "REFTensoRF":
        total = total + 0.5 * m.penalty
        m.penalty = torch.zeros((), device="cuda")
    total.backward()
    if bucket: bucket.all_reduce_mean()
    if int(os.environ.get("TVR_LOOP_SYNC", "0")) == 2: opt.found_inf = m.training_fault_flag()
    opt.step()
    return loss
N = 20
if GRAPH:
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    if os.environ.get("TVR_DEBUG_HDR"):
        import ctypes as C
        from jittor_myc_nerfs_amd import _lib as L
        def hdr():
            b = m._train_buf
            lay = L.ScratchLayout(); L.check(L.lib().tvr_scratch_describe(b["key"][0], b["key"][1], C.byref(lay)), "d")
            torch.cuda.synchronize()
            return b["key"], b["scratch"].data_ptr(), b["scratch"][lay.counter:lay.counter + 32].view(torch.int32).tolist()
        print("after warm-up", hdr(), flush=True)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        l = step()
    if os.environ.get("TVR_DEBUG_HDR"):
        print("after capture", hdr(), flush=True)
        for i in range(4):
            graph.replay(); print("replay", i, hdr(), flush=True)
    for _ in range(10): graph.replay()
    dt = 1e9
    for _blk in range(3):                             # fastest of three blocks: the first block of a fresh process on a fresh box has read 4.2 - 4.6 ms for 3.4 (round 6)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(N): graph.replay()
        torch.cuda.synchronize(); dt = min(dt, (time.perf_counter() - t0) / N)
    fault = m.check_training_faults()                 # (None on a healthy run; 'overflow' means the timed steps worked on a truncated queue)
    if fault is not None: print("WARNING: check_training_faults() ->", fault)
else:
    SYNC = int(os.environ.get("TVR_LOOP_SYNC", "0"))          # 1: read the loss and the fault flags on the host after every step, as train.py:262 does;
                                                              # 2: reconstruct.py's loop — the flags guard the fused Adam on the device, no host read
    for _ in range(3): step()
    dt = 1e9
    for _blk in range(3):                             # fastest of three blocks (a host-paced step on a shared host)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(N):
            l = step()
            if SYNC == 1:
                m.check_training_faults(); float(l.detach())
        torch.cuda.synchronize(); dt = min(dt, (time.perf_counter() - t0) / N)
if rank == 0: print(f"{MODEL} (fused MLP kernels {int(m.fused_mlp_training)}, static step {int(m.static_training)}, hipGraph {int(GRAPH)}) train step ({world} rank(s), {4096 // world} rays each): {dt * 1e3:.2f} ms  ({1 / dt:.1f} it/s), batch 4096 rays x {nS} samples, loss {float(l.detach()):.3e}")
