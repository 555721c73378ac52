"""Output helpers with the reference's file formats (src/liftreg/utils/utils.py)."""
import os

import numpy as np


def write_nifti1_gz(path, data, affine=None):
    """A single-file NIfTI-1 image (`.nii.gz`), what `nib.save(nib.Nifti1Image(data, np.eye(4)), path)` produces at
    utils/utils.py:66-67: 348-byte header, 4 zero extension bytes, voxels in Fortran order (first index fastest),
    float32/float64/int16/uint8, sform = `affine` (code 2, "aligned"), no qform, no scaling.  nibabel / SimpleITK are not
    installed in this image, so the writer is pinned to the published NIfTI-1 layout (and to its own reader,
    `read_nifti1_gz`) — NOT byte-compared with a nibabel file: readers ignore the fields where the two could differ."""
    import gzip
    import struct
    data = np.asarray(data)
    codes = {np.dtype(np.float32): (16, 32), np.dtype(np.float64): (64, 64), np.dtype(np.int16): (4, 16), np.dtype(np.uint8): (2, 8)}
    if data.dtype not in codes or data.ndim < 1 or data.ndim > 7:
        raise ValueError(f"unsupported array for NIfTI-1: dtype {data.dtype}, ndim {data.ndim}")
    code, bitpix = codes[data.dtype]
    aff = np.eye(4) if affine is None else np.asarray(affine, dtype=np.float64)
    dim = [data.ndim] + list(data.shape) + [1] * (7 - data.ndim)
    pixdim = [1.0] + [float(np.linalg.norm(aff[:3, i])) if i < 3 else 1.0 for i in range(min(data.ndim, 7))] + [1.0] * (7 - data.ndim)
    hdr = bytearray(348)
    struct.pack_into("<i", hdr, 0, 348)                      # sizeof_hdr
    struct.pack_into("<8h", hdr, 40, *dim)                   # dim[8]
    struct.pack_into("<hh", hdr, 70, code, bitpix)           # datatype, bitpix
    struct.pack_into("<8f", hdr, 76, *pixdim[:8])            # pixdim[8] (pixdim[0] = qfac)
    struct.pack_into("<f", hdr, 108, 352.0)                  # vox_offset
    struct.pack_into("<f", hdr, 112, float("nan"))           # scl_slope: nibabel writes NaN for "no scaling"
    struct.pack_into("<f", hdr, 116, float("nan"))           # scl_inter
    struct.pack_into("<hh", hdr, 252, 0, 2)                  # qform_code = 0 (unknown), sform_code = 2 (aligned)
    struct.pack_into("<4f", hdr, 280, *aff[0])               # srow_x
    struct.pack_into("<4f", hdr, 296, *aff[1])               # srow_y
    struct.pack_into("<4f", hdr, 312, *aff[2])               # srow_z
    hdr[344:348] = b"n+1\x00"                                # magic: single file
    with gzip.open(path, "wb", compresslevel=1) as fh:
        fh.write(bytes(hdr))
        fh.write(b"\x00\x00\x00\x00")                         # no header extensions
        fh.write(np.asfortranarray(data).tobytes(order="F"))


def read_nifti1_gz(path):
    """(array, sform affine) of a single-file NIfTI-1 image written by `write_nifti1_gz` (or any little-endian one)."""
    import gzip
    import struct
    with gzip.open(path, "rb") as fh:
        raw = fh.read()
    if struct.unpack_from("<i", raw, 0)[0] != 348 or raw[344:347] != b"n+1":
        raise ValueError("not a little-endian single-file NIfTI-1 image")
    dim = struct.unpack_from("<8h", raw, 40)
    code = struct.unpack_from("<h", raw, 70)[0]
    dt = {16: np.float32, 64: np.float64, 4: np.int16, 2: np.uint8}[code]
    off = int(struct.unpack_from("<f", raw, 108)[0])
    shape = tuple(dim[1:1 + dim[0]])
    arr = np.frombuffer(raw, dtype=dt, count=int(np.prod(shape)), offset=off).reshape(shape, order="F")
    aff = np.eye(4)
    aff[0], aff[1], aff[2] = (struct.unpack_from("<4f", raw, o) for o in (280, 296, 312))
    return arr, aff


def save_deformations(phis, fname_list, output_path, nifti=True):
    """`{id}_phi.npy` and `{id}_phi.nii.gz` = (phi+1)/2 as float32 (B,3,D,W,H → one file pair per sample),
    utils/utils.py:57-68.  The `.npy` is what the evaluation scripts read (tools/evaluate_dir_lab.py:176-178); the
    `.nii.gz` (identity affine, as the reference's `nib.Nifti1Image(phis[i], np.eye(4))`) is written by `write_nifti1_gz`.
    """
    phis = (phis.detach().cpu().numpy() if hasattr(phis, "detach") else np.asarray(phis))
    phis = (phis + 1.) / 2.
    for i, name in enumerate(fname_list):
        arr = phis[i].astype(np.float32)
        if nifti:
            write_nifti1_gz(os.path.join(output_path, f"{name}_phi.nii.gz"), arr, np.eye(4))
        np.save(os.path.join(output_path, f"{name}_phi.npy"), arr)


def sigmoid_decay(ep, static=5, k=5):
    """factor = k/(k+exp((ep-static)/k)) after `static` epochs, 1 before (utils/utils.py:93-107)."""
    if ep < static:
        return float(1.)
    ep = ep - static
    return float(k / (k + np.exp(ep / k)))


def compute_jacobi_map(map, spacing, crop_boundary=True, use_01=False):
    """(mean over the batch of Σ|det J| where det J < 0, mean number of folded voxels) — utils/utils.py:20-55.

    Like the reference, the cropped statistics are computed-and-discarded there (`:46-52` overwrite them), so the
    values returned are over the whole volume.  One pass on the GPU (`lr_jacobi_det_stats_f32`); the derivative
    stencil is the assumed mermaid one (parity unpinned, see csrc/metrics.hip)."""
    import torch

    from .. import ops
    t = map if isinstance(map, torch.Tensor) else torch.from_numpy(np.asarray(map, dtype=np.float32))
    if not t.is_cuda:
        t = t.to("cuda")
    span = 1.0 if use_01 else 2.0
    sp = np.asarray(spacing, dtype=np.float64) * span
    abs_sum, count = ops.jacobi_det_stats(t.float().contiguous(), sp).tolist()
    return abs_sum / t.shape[0], count / t.shape[0]
