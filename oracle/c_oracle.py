"""oracle/c_oracle.py — TEST INFRASTRUCTURE ONLY.

ctypes/numpy front-end of oracle/liftreg_oracle.c (plain-C restatement of the
LiftReg hot path).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this; liftreg_amd never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")


def build(force=False):
    """Compile the C oracle (gcc) if it is not there yet."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(
            os.path.join(_HERE, "liftreg_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def _chk(rc):
    if rc != 0:
        raise RuntimeError(f"oracle error {rc}")


def calc_relative_atten_coef(hu):
    hu, ph = _f(hu)
    out = np.empty_like(hu)
    lib().or_calc_relative_atten_coef(ph, out.ctypes.data_as(C.POINTER(C.c_float)), C.c_int64(hu.size))
    return out


def drr_sample_coords(poses, spacing, shape, resolution):
    D, W, H = shape
    Rd, Rh = resolution
    poses, pp = _f(poses)
    sp, psp = _f(spacing)
    P = poses.shape[0]
    pix = np.empty((P, Rd, Rh, W, 3), np.float32)
    dx = np.empty((P, Rd, Rh), np.float32)
    _chk(lib().or_drr_sample_coords_f32(pp, psp, pix.ctypes.data_as(C.POINTER(C.c_float)),
                                        dx.ctypes.data_as(C.POINTER(C.c_float)), D, W, H, P, Rd, Rh))
    return pix, dx


def drr_forward(vol, poses, spacing, resolution, d0=0, d1=None, full_D=None, flags=0):
    vol, pv = _f(vol)
    Ds, W, H = vol.shape
    D = full_D if full_D is not None else Ds
    d1 = D if d1 is None else d1
    assert d1 - d0 == Ds
    poses, pp = _f(poses)
    sp, psp = _f(spacing)
    P = poses.shape[0]
    Rd, Rh = resolution
    out = np.empty((P, Rd, Rh), np.float32)
    _chk(lib().or_drr_forward_f32(pv, pp, psp, out.ctypes.data_as(C.POINTER(C.c_float)), D, W, H, d0,
                                  d1, P, Rd, Rh, flags))
    return out


def backproject_coords(poses, shape, proj_shape):
    D, W, H = shape
    Pw, Ph = proj_shape
    poses, pp = _f(poses)
    P = poses.shape[0]
    pix = np.empty((P, D, W, H, 2), np.float32)
    _chk(lib().or_backproject_coords_f32(pp, pix.ctypes.data_as(C.POINTER(C.c_float)), P, Pw, Ph, D, W, H))
    return pix


def backproject(proj, poses, shape, d0=0, d1=None):
    proj, ppj = _f(proj)
    B, P, Pw, Ph = proj.shape
    D, W, H = shape
    d1 = D if d1 is None else d1
    poses, pp = _f(poses)
    out = np.empty((B, P, d1 - d0, W, H), np.float32)
    _chk(lib().or_backproject_f32(ppj, pp, out.ctypes.data_as(C.POINTER(C.c_float)), B, P, Pw, Ph, D, W,
                                  H, d0, d1, C.c_int64(out[0].size)))
    return out


def conv3d_k3_lrelu(x, w, b, stride, slope=0.2):
    x, px = _f(x)
    w, pw = _f(w)
    B, Cin, D, W, H = x.shape
    Cout = w.shape[0]
    pb = None
    if b is not None:
        b, pb = _f(b)
    o = lambda n: (n - 1) // stride + 1
    out = np.empty((B, Cout, o(D), o(W), o(H)), np.float32)
    _chk(lib().or_conv3d_k3_lrelu_f32(px, pw, pb, out.ctypes.data_as(C.POINTER(C.c_float)), B, Cin, Cout,
                                      D, W, H, stride, C.c_float(slope)))
    return out


def linear_lrelu(x, w, b, slope=1.0):
    x, px = _f(x)
    w, pw = _f(w)
    pb = None
    if b is not None:
        b, pb = _f(b)
    B, K = x.shape
    O = w.shape[0]
    y = np.empty((B, O), np.float32)
    _chk(lib().or_linear_lrelu_f32(px, pw, pb, y.ctypes.data_as(C.POINTER(C.c_float)), B, K, O,
                                   C.c_float(slope)))
    return y


def pca_reconstruct(coefs, basis, mean):
    coefs, pc = _f(coefs)
    basis, pb = _f(basis)
    mean, pm = _f(mean)
    B, L = coefs.shape
    M = basis.shape[1]
    disp = np.empty((B, M), np.float32)
    _chk(lib().or_pca_reconstruct_f32(pc, pb, pm, disp.ctypes.data_as(C.POINTER(C.c_float)), B, L,
                                      C.c_int64(M), C.c_int64(M), C.c_int64(M)))
    return disp


USING_SCALE, BORDER, NEAREST = 1, 2, 4


def warp(img, disp, ids=None, seg=None, flags=USING_SCALE, d0=0, d1=None):
    """Returns (phi, warped).  ids = (id0, id1, id2) per-axis identity tables or None."""
    img, pi = _f(img)
    disp, pd = _f(disp)
    B, Cc, D, W, H = img.shape
    d1 = D if d1 is None else d1
    Dn = d1 - d0
    assert disp.shape == (B, 3, Dn, W, H)
    ps = None
    if seg is not None:
        seg, ps = _f(seg)
    p0 = p1 = p2 = None
    if ids is not None:
        i0, p0 = _f(ids[0])
        i1, p1 = _f(ids[1])
        i2, p2 = _f(ids[2])
    phi = np.empty_like(disp)
    warped = np.empty((B, Cc, Dn, W, H), np.float32)
    _chk(lib().or_warp_trilinear_f32(pi, ps, pd, p0, p1, p2, phi.ctypes.data_as(C.POINTER(C.c_float)),
                                     warped.ctypes.data_as(C.POINTER(C.c_float)), B, Cc, D, W, H, d0, d1,
                                     flags))
    return phi, warped


def mask_compose(img, seg):
    img, pi = _f(img)
    seg, ps = _f(seg)
    out = np.empty_like(img)
    lib().or_mask_compose_f32(pi, ps, out.ctypes.data_as(C.POINTER(C.c_float)), C.c_int64(img.size))
    return out


def ncc_loss(x, y, variant=0):
    """variant 0: layers/losses.py NCCLoss on (B, -1); variant 1: layers/layers.py NCCLoss on (B*C, -1)."""
    x = np.ascontiguousarray(x, np.float32)
    y = np.ascontiguousarray(y, np.float32)
    n_batch = x.shape[0]
    R = n_batch if variant == 0 else x.shape[0] * x.shape[1]
    N = x.size // R
    loss = np.zeros(1, np.float32)
    rows = np.zeros(R, np.float32)
    _chk(lib().or_ncc_loss_f32(x.ctypes.data_as(C.POINTER(C.c_float)), y.ctypes.data_as(C.POINTER(C.c_float)),
                               loss.ctypes.data_as(C.POINTER(C.c_float)),
                               rows.ctypes.data_as(C.POINTER(C.c_float)), R, C.c_int64(N), n_batch, variant))
    return float(loss[0]), rows


def ncc_moments(x, y, R):
    x = np.ascontiguousarray(x, np.float32)
    y = np.ascontiguousarray(y, np.float32)
    N = x.size // R
    m = np.zeros((R, 5), np.float64)
    lib().or_ncc_moments_f32(x.ctypes.data_as(C.POINTER(C.c_float)), y.ctypes.data_as(C.POINTER(C.c_float)),
                             m.ctypes.data_as(C.POINTER(C.c_double)), R, C.c_int64(N))
    return m


def fastdiv_mismatches(d, lo=2.0 ** -20, hi=2.0 ** 12):
    """How many floats x, lo <= |x| < hi, have x / d != the projector's reciprocal division (drr_forward.hip: div_by)."""
    import struct
    f = lib().or_fastdiv_mismatches
    f.restype = C.c_int64
    f.argtypes = [C.c_float, C.c_uint32, C.c_uint32]
    bits = lambda v: struct.unpack("I", struct.pack("f", v))[0]
    return int(f(float(d), bits(lo), bits(hi)))
