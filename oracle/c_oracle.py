"""ctypes binding of oracle/tvr_oracle.c (scalar-C oracle (b)).  TEST INFRASTRUCTURE ONLY —
see tvr_oracle.c / tensorf_oracle.py headers.  Built by oracle/Makefile into oracle/_build/."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libtvr_oracle.so")
_FP = C.POINTER(C.c_float)


class _Scene(C.Structure):
    _fields_ = [("aabb", C.c_float * 6), ("grid", C.c_int32 * 3), ("cd", C.c_int32 * 3), ("ca", C.c_int32 * 3),
                ("app_dim", C.c_int32), ("featC", C.c_int32), ("view_pe", C.c_int32), ("fea_pe", C.c_int32),
                ("dplane", _FP * 3), ("dline", _FP * 3), ("aplane", _FP * 3), ("aline", _FP * 3),
                ("basis", _FP), ("W1", _FP), ("b1", _FP), ("W2", _FP), ("b2", _FP), ("W3", _FP), ("b3", _FP),
                ("near_", C.c_float), ("far_", C.c_float), ("step", C.c_float), ("density_shift", C.c_float),
                ("distance_scale", C.c_float), ("thres", C.c_float), ("act", C.c_int32),
                ("alpha_vol", _FP), ("agrid", C.c_int32 * 3), ("alpha_aabb", C.c_float * 6)]


class _Dump(C.Structure):
    _fields_ = [("z", _FP), ("valid", C.POINTER(C.c_uint8)), ("bbox_valid", C.POINTER(C.c_uint8)),
                ("cell", C.POINTER(C.c_int32)), ("sf", _FP), ("sigma", _FP), ("alpha", _FP), ("weight", _FP),
                ("app", C.POINTER(C.c_uint8)), ("rgb", _FP), ("tmin", _FP), ("acc", _FP)]


def build(force: bool = False) -> str:
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "tvr_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.tvr_oracle_render.restype = C.c_int
        _lib.tvr_oracle_render.argtypes = [C.POINTER(_Scene), _FP, C.c_int64, C.c_int, C.c_int, _FP, _FP, _FP,
                                           C.POINTER(_Dump), C.c_int]
        for n in ("tvr_oracle_density_features", "tvr_oracle_app_features", "tvr_oracle_alpha_samples"):
            getattr(_lib, n).restype = None
            getattr(_lib, n).argtypes = [C.POINTER(_Scene), _FP, C.c_int64, _FP]
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(_FP)


class COracle:
    """Holds fp32 copies of a scene (reference layout) and exposes the scalar-C oracle."""

    def __init__(self, arrs: Dict[str, np.ndarray], step: float, near_far, density_shift=-10.0, distance_scale=25.0,
                 rayMarch_weight_thres=1e-4, fea2denseAct="softplus", view_pe=2, fea_pe=2, **_ignored):
        self.keep = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in arrs.items() if k != "gridSize"}
        k = self.keep
        s = _Scene()
        s.aabb[:] = list(k["aabb"].reshape(-1))
        g = [int(x) for x in arrs["gridSize"]]
        s.grid[:] = g
        for i in range(3):
            s.cd[i] = k[f"density_plane.{i}"].shape[1]
            s.ca[i] = k[f"app_plane.{i}"].shape[1]
            s.dplane[i], s.dline[i] = _p(k[f"density_plane.{i}"]), _p(k[f"density_line.{i}"])
            s.aplane[i], s.aline[i] = _p(k[f"app_plane.{i}"]), _p(k[f"app_line.{i}"])
        s.app_dim, s.featC = k["basis_mat"].shape[0], k["W1"].shape[0]
        s.view_pe, s.fea_pe = view_pe, fea_pe
        s.basis = _p(k["basis_mat"])
        for n in ("W1", "b1", "W2", "b2", "W3", "b3"):
            setattr(s, n, _p(k[n]))
        s.near_, s.far_ = float(near_far[0]), float(near_far[1])
        s.step = float(step)
        s.density_shift, s.distance_scale, s.thres = density_shift, distance_scale, rayMarch_weight_thres
        s.act = 0 if fea2denseAct == "softplus" else 1
        if "alpha_volume" in k:
            v = k["alpha_volume"]
            s.alpha_vol = _p(v)
            s.agrid[:] = [v.shape[-1], v.shape[-2], v.shape[-3]]
            s.alpha_aabb[:] = list(k["alpha_aabb"].reshape(-1))
        self.s = s
        self.app_dim = int(s.app_dim)

    def render(self, rays: np.ndarray, S: int, white_bg=True, jitter: Optional[np.ndarray] = None, dump=False, nthreads=1):
        rays = np.ascontiguousarray(rays, np.float32)
        n = rays.shape[0]
        rgb, depth = np.zeros((n, 3), np.float32), np.zeros((n,), np.float32)
        d, out = None, {}
        if dump:
            d = _Dump()
            out = dict(z=np.zeros((n, S), np.float32), valid=np.zeros((n, S), np.uint8), bbox_valid=np.zeros((n, S), np.uint8),
                       cell=np.zeros((n, S, 3), np.int32), sf=np.zeros((n, S), np.float32), sigma=np.zeros((n, S), np.float32),
                       alpha=np.zeros((n, S), np.float32), weight=np.zeros((n, S), np.float32), app=np.zeros((n, S), np.uint8),
                       rgb=np.zeros((n, S, 3), np.float32), tmin=np.zeros((n,), np.float32), acc=np.zeros((n,), np.float32))
            for name, a in out.items():
                setattr(d, name, a.ctypes.data_as(dict(_Dump._fields_)[name]))
        jit = None if jitter is None else np.ascontiguousarray(jitter, np.float32)
        rc = lib().tvr_oracle_render(C.byref(self.s), _p(rays), n, int(S), int(bool(white_bg)),
                                     None if jit is None else _p(jit), _p(rgb), _p(depth),
                                     None if d is None else C.byref(d), int(nthreads))
        assert rc == 0
        out.update(rgb_map=rgb, depth_map=depth)
        return out

    def density_features(self, xyz_norm: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(xyz_norm, np.float32)
        o = np.zeros((x.shape[0],), np.float32)
        lib().tvr_oracle_density_features(C.byref(self.s), _p(x), x.shape[0], _p(o))
        return o

    def app_features(self, xyz_norm: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(xyz_norm, np.float32)
        o = np.zeros((x.shape[0], self.app_dim), np.float32)
        lib().tvr_oracle_app_features(C.byref(self.s), _p(x), x.shape[0], _p(o))
        return o

    def alpha_samples(self, xyz: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(xyz, np.float32)
        o = np.zeros((x.shape[0],), np.float32)
        lib().tvr_oracle_alpha_samples(C.byref(self.s), _p(x), x.shape[0], _p(o))
        return o
