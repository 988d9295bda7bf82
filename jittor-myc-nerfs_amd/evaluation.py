"""Callers and data formats either side of the render path: Blender-format datasets in, PNG + metrics out.

  * BlenderRays            — the ray side of `BlenderDataset` (tensorf-myc/dataLoader/blender.py:13-161): reads `transforms_{split}.json`,
                             builds `all_rays [n_img, H*W, 6]`, loads ground-truth PNGs when they exist (alpha blended to white, :105-108)
  * evaluation             — tensorf-myc/renderer.py:29-91: per-image render (chunk 1024), clamp, PSNR / SSIM, `*_r_{idx}.png`, `rgbd/*.png`, `mean.txt`
  * evaluation_path        — tensorf-myc/renderer.py:93-148 without the mp4 writers (imageio is not available here)
  * rgb_ssim, visualize_depth_numpy — tensorf-myc/utils.py:73-119, :11-26 (the depth colour map is a numpy jet ramp standing in for cv2's LUT)
Everything here is host-side plumbing around `renderer(rays, tensorf, …)`; the rendering itself is the HIP path.
"""
from __future__ import annotations

import json
import os
from typing import List, Optional

import numpy as np
import torch

from . import rays as R


class BlenderRays:
    def __init__(self, datadir: str, split: str = "test", downsample: float = 1.0, is_stack: bool = True, N_vis: int = -1,
                 near: Optional[float] = None, far: Optional[float] = None, bbox=None, white_bg: bool = True):
        self.root_dir, self.split, self.is_stack, self.white_bg = datadir, split, is_stack, white_bg
        self.img_wh = (int(800 / downsample), int(800 / downsample))                                    # blender.py:19
        self.near_far = [5.0, 40.0] if near is None or far is None else [near, far]                     # blender.py:50-53
        self.scene_bbox = torch.tensor([[-5.0] * 3, [5.0] * 3] if bbox is None else bbox, dtype=torch.float32).view(2, 3)
        name = f"transforms_{split}.json"
        path = os.path.join(datadir, name)
        if not os.path.exists(path) and split == "test":            # data_refine/Easyship ships it as transform_test.json
            path = os.path.join(datadir, "transform_test.json")
        with open(path) as f:
            self.meta = json.load(f)
        w, h = self.img_wh
        self.camera_angle_x = float(self.meta["camera_angle_x"])
        self.focal = R.focal_from_angle(self.camera_angle_x, w)
        frames = self.meta["frames"]
        interval = 1 if N_vis < 0 else max(len(frames) // N_vis, 1)
        self.poses, rays, rgbs = [], [], []
        for fr in frames[::interval]:
            M = np.asarray(fr["transform_matrix"], dtype=np.float64)
            self.poses.append(torch.tensor(M @ R.BLENDER2OPENCV, dtype=torch.float32))
            rays.append(R.frame_rays(M, h, w, self.camera_angle_x))
            img_path = os.path.join(datadir, f"{fr['file_path']}.png")
            if os.path.exists(img_path):
                from PIL import Image
                img = Image.open(img_path)
                if img.size != (w, h):
                    img = img.resize((w, h), Image.LANCZOS)
                a = (np.asarray(img).astype(np.float32) / 255.0).reshape(h * w, -1)
                if a.shape[1] == 4:
                    a = a[:, :3] * a[:, 3:] + (1 - a[:, 3:])                                              # blender.py:108
                rgbs.append(torch.from_numpy(np.ascontiguousarray(a[:, :3])))
        self.all_rays = torch.stack(rays, 0) if is_stack else torch.cat(rays, 0)
        self.all_rgbs = (torch.stack(rgbs, 0).reshape(-1, h, w, 3) if is_stack else torch.cat(rgbs, 0)) if len(rgbs) == len(rays) else []


def rgb_ssim(img0, img1, max_val, filter_size=11, filter_sigma=1.5, k1=0.01, k2=0.03):
    """utils.py:73-119 (mip-NeRF SSIM): separable 11-tap Gaussian, 'valid' convolution, per channel."""
    import scipy.signal
    img0, img1 = np.asarray(img0, np.float64), np.asarray(img1, np.float64)
    hw = filter_size // 2
    shift = (2 * hw - filter_size + 1) / 2
    filt = np.exp(-0.5 * ((np.arange(filter_size) - hw + shift) / filter_sigma) ** 2)
    filt /= filt.sum()
    conv = lambda z, f: scipy.signal.convolve2d(z, f, mode="valid")
    blur = lambda z: np.stack([conv(conv(z[..., i], filt[:, None]), filt[None, :]) for i in range(z.shape[-1])], -1)
    mu0, mu1 = blur(img0), blur(img1)
    mu00, mu11, mu01 = mu0 * mu0, mu1 * mu1, mu0 * mu1
    s00 = np.maximum(0.0, blur(img0 ** 2) - mu00)
    s11 = np.maximum(0.0, blur(img1 ** 2) - mu11)
    s01 = blur(img0 * img1) - mu01
    s01 = np.sign(s01) * np.minimum(np.sqrt(s00 * s11), np.abs(s01))
    c1, c2 = (k1 * max_val) ** 2, (k2 * max_val) ** 2
    return float(np.mean(((2 * mu01 + c1) * (2 * s01 + c2)) / ((mu00 + mu11 + c1) * (s00 + s11 + c2))))


def rgb_ssim_torch(img0: torch.Tensor, img1: torch.Tensor, max_val, filter_size=11, filter_sigma=1.5, k1=0.01, k2=0.03) -> float:
    """`rgb_ssim` on the tensors' device: the same separable 11-tap Gaussian, 'valid' window, float64 — two depthwise conv2d calls per
    blur instead of 30 scipy convolutions per frame (0.4 s per 800x800 frame on the host, the render itself takes 23 ms)."""
    return float(_rgb_ssim_dev(img0, img1, max_val, filter_size, filter_sigma, k1, k2))


def _rgb_ssim_dev(img0, img1, max_val, filter_size=11, filter_sigma=1.5, k1=0.01, k2=0.03) -> torch.Tensor:
    """The 0-dim result of rgb_ssim_torch, left on the device (no host synchronisation: the evaluation loop fetches it with the frame)."""
    x0 = img0.to(torch.float64).permute(2, 0, 1).unsqueeze(1)          # [3,1,H,W]: channels as a batch
    x1 = img1.to(device=x0.device, dtype=torch.float64).permute(2, 0, 1).unsqueeze(1)
    hw = filter_size // 2
    shift = (2 * hw - filter_size + 1) / 2
    f = torch.exp(-0.5 * ((torch.arange(filter_size, dtype=torch.float64, device=x0.device) - hw + shift) / filter_sigma) ** 2)
    f = f / f.sum()
    blur = lambda z: torch.nn.functional.conv2d(torch.nn.functional.conv2d(z, f.view(1, 1, -1, 1)), f.view(1, 1, 1, -1))
    mu0, mu1 = blur(x0), blur(x1)
    mu00, mu11, mu01 = mu0 * mu0, mu1 * mu1, mu0 * mu1
    s00 = torch.clamp(blur(x0 * x0) - mu00, min=0.0)
    s11 = torch.clamp(blur(x1 * x1) - mu11, min=0.0)
    s01 = blur(x0 * x1) - mu01
    s01 = torch.sign(s01) * torch.minimum(torch.sqrt(s00 * s11), torch.abs(s01))
    c1, c2 = (k1 * max_val) ** 2, (k2 * max_val) ** 2
    return torch.mean(((2 * mu01 + c1) * (2 * s01 + c2)) / ((mu00 + mu11 + c1) * (s00 + s11 + c2)))


class _ImageWriter:
    """PNG encoding off the render loop: a few worker threads (zlib releases the GIL); `close()` waits for the files."""

    def __init__(self, workers: int = 0):
        from concurrent.futures import ThreadPoolExecutor
        workers = workers or max(4, min(12, (os.cpu_count() or 8) // 2))
        self.pool, self.jobs = ThreadPoolExecutor(max_workers=workers), []

    def write(self, path, arr):
        self.jobs.append(self.pool.submit(_imwrite, path, arr))

    def close(self):
        for j in self.jobs:
            j.result()
        self.pool.shutdown()


_PNG_LEVEL = int(os.environ.get("TVR_PNG_LEVEL", "1"))
_LOOKAHEAD = os.environ.get("TVR_EVAL_LOOKAHEAD", "1") != "0"     # 0: finish every frame on the host before the next is enqueued (the A/B of scripts/eval_loop_timing.py)


class _FrameFetch:
    """Device -> host copies of a finished frame on a SIDE stream into pinned memory, so that the loop can enqueue frame k + 1 before it touches frame k on the host
    (round 5: a plain `.cpu()` is ordered behind everything queued on the render stream, the next frame included; the reference synchronises per 1024-ray chunk,
    renderer.py:23-25).  On a CPU device `start` is a plain conversion."""

    def __init__(self, device):
        self.cuda = torch.device(device).type == "cuda" and torch.cuda.is_available()
        self.stream = torch.cuda.Stream(device) if self.cuda else None

    def start(self, *tensors):
        if not self.cuda:
            return [t.detach().cpu().numpy() for t in tensors], None
        ready = torch.cuda.Event()
        ready.record()                                              # behind the frame's kernels (and metrics) on the render stream
        host = []
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ready)
            for t in tensors:
                h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                h.copy_(t, non_blocking=True)
                t.record_stream(self.stream)
                host.append(h)
            done = torch.cuda.Event()
            done.record(self.stream)
        return host, done

    @staticmethod
    def finish(handle):
        host, done = handle
        if done is None:
            return host
        done.synchronize()
        return [h.numpy() for h in host]


def visualize_depth_numpy(depth, minmax=None):
    """utils.py:11-26 with a numpy jet ramp instead of cv2.applyColorMap (returns BGR-ordered uint8 like cv2 does)."""
    x = np.nan_to_num(depth)
    mi, ma = (np.min(x[x > 0]) if (x > 0).any() else 0.0, np.max(x)) if minmax is None else minmax
    x = np.clip((x - mi) / (ma - mi + 1e-8), 0, 1)
    x = (255 * x).astype(np.uint8).astype(np.float32) / 255.0
    r = np.clip(1.5 - np.abs(4 * x - 3), 0, 1)
    g = np.clip(1.5 - np.abs(4 * x - 2), 0, 1)
    b = np.clip(1.5 - np.abs(4 * x - 1), 0, 1)
    return (np.stack([b, g, r], -1) * 255).astype(np.uint8), [mi, ma]


def _imwrite(path, arr):
    from PIL import Image
    # compress_level 1: the same pixels (PNG is lossless), files ~ 15 % larger, encoding 3 - 4 x faster than Pillow's default 6 — the encoder, not the renderer, sets the pace of
    # an evaluation loop (scripts/eval_loop_timing.py: 48 ms per 800 x 800 frame with the default level against 20 ms of kernels)
    Image.fromarray(arr).save(path, compress_level=_PNG_LEVEL)


@torch.no_grad()
def evaluation(test_dataset, tensorf, args, renderer, savePath=None, N_vis=5, prtx='', N_samples=-1, white_bg=False, ndc_ray=False,
               compute_extra_metrics=True, device='cuda') -> List[float]:
    PSNRs, ssims = [], []
    if savePath is not None:
        os.makedirs(savePath + "/rgbd", exist_ok=True)
    near_far = test_dataset.near_far
    n = test_dataset.all_rays.shape[0]
    interval = 1 if N_vis < 0 else max(n // N_vis, 1)
    idxs = list(range(0, n, interval))
    W, H = test_dataset.img_wh
    expname = getattr(args, "expname", "render") if args is not None else "render"
    writer = _ImageWriter() if savePath is not None else None
    fetch = _FrameFetch(device)

    def finish(idx, handle, n_metrics):
        arrs = fetch.finish(handle)
        rgb, depth, metrics = arrs[0], arrs[1], arrs[2:]
        if n_metrics >= 1:
            PSNRs.append(-10.0 * np.log(float(metrics[0])) / np.log(10.0))
        if n_metrics >= 2:
            ssims.append(float(metrics[1]))
        img = (rgb * 255).astype('uint8')
        if savePath is not None:
            depth_vis, _ = visualize_depth_numpy(depth, near_far)
            writer.write(f'{savePath}/{expname}_r_{idx}.png', img)
            writer.write(f'{savePath}/rgbd/{prtx}{idx:03d}.png', np.concatenate((img, depth_vis), axis=1))

    pending = None                                                   # one frame of lookahead: frame k is post-processed on the host while frame k + 1 renders
    for idx, samples in enumerate(test_dataset.all_rays[0::interval]):
        rays = samples.view(-1, samples.shape[-1]).to(device)
        rgb_map, _, depth_map, _, _ = renderer(rays, tensorf, chunk=1024, N_samples=N_samples, ndc_ray=ndc_ray, white_bg=white_bg, device=device)
        rgb_dev = rgb_map.clamp(0.0, 1.0).reshape(H, W, 3)
        metrics = []
        if len(test_dataset.all_rgbs):
            gt = test_dataset.all_rgbs[idxs[idx]].view(H, W, 3).to(rgb_dev.device)
            metrics.append(torch.mean((rgb_dev - gt) ** 2))
            if compute_extra_metrics:
                metrics.append(_rgb_ssim_dev(rgb_dev, gt, 1))                # utils.py:73-119 on the device
        handle = fetch.start(rgb_dev, depth_map.reshape(H, W), *metrics)
        if pending is not None:
            finish(*pending)
        pending = (idx, handle, len(metrics))
        if not _LOOKAHEAD:
            finish(*pending)
            pending = None
    if pending is not None:
        finish(*pending)
    if writer is not None:
        writer.close()
    if PSNRs and savePath is not None:
        vals = [np.mean(PSNRs)] + ([np.mean(ssims), 0.0, 0.0] if compute_extra_metrics else [])
        np.savetxt(f'{savePath}/{prtx}mean.txt', np.asarray(vals))
    return PSNRs


@torch.no_grad()
def evaluation_path(test_dataset, tensorf, c2ws, renderer, savePath=None, N_vis=5, prtx='', N_samples=-1, white_bg=False, ndc_ray=False,
                    compute_extra_metrics=True, device='cuda'):
    if savePath is not None:
        os.makedirs(savePath + "/rgbd", exist_ok=True)
    W, H = test_dataset.img_wh
    focal = test_dataset.focal
    dirs = R.get_ray_directions(H, W, [focal, focal])
    dirs = dirs / np.sqrt((dirs * dirs).sum(-1, keepdims=True))
    frames = []
    writer = _ImageWriter() if savePath is not None else None
    fetch = _FrameFetch(device)

    def finish(idx, handle):
        rgb, depth = fetch.finish(handle)
        img = (rgb * 255).astype('uint8')
        frames.append(img)
        if savePath is not None:
            depth_vis, _ = visualize_depth_numpy(depth, test_dataset.near_far)
            writer.write(f'{savePath}/{prtx}{idx:03d}.png', img)
            writer.write(f'{savePath}/rgbd/{prtx}{idx:03d}.png', np.concatenate((img, depth_vis), axis=1))

    pending = None
    for idx, c2w in enumerate(c2ws):
        o, d = R.get_rays(dirs, np.asarray(c2w, dtype=np.float32))
        rays = torch.from_numpy(np.ascontiguousarray(np.concatenate([o, d], 1), dtype=np.float32)).to(device)
        rgb_map, _, depth_map, _, _ = renderer(rays, tensorf, chunk=8192, N_samples=N_samples, ndc_ray=ndc_ray, white_bg=white_bg, device=device)
        handle = fetch.start(rgb_map.clamp(0.0, 1.0).reshape(H, W, 3), depth_map.reshape(H, W))
        if pending is not None:
            finish(*pending)
        pending = (idx, handle)
        if not _LOOKAHEAD:
            finish(*pending)
            pending = None
    if pending is not None:
        finish(*pending)
    if writer is not None:
        writer.close()
    return frames
