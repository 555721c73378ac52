"""ctypes binding of libliftreg_hip.so (the C ABI declared in include/liftreg_hip.h).

There is NO CPU or PyTorch fallback: if the library is missing or was built for
another target, importing the ops raises.  The library is built in-tree by
`__graft_entry__.build()` / `make -C liftreg_amd/csrc` (hipcc --offload-arch=gfx950).
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# LIFTREG_HIP_LIB: another build of the SAME library (the experimental one, `make -C liftreg_amd/csrc exp`; A/B variants)
LIB_PATH = os.environ.get("LIFTREG_HIP_LIB") or os.path.join(CSRC, "libliftreg_hip.so")

LR_OK = 0
LAYOUT_NCDHW, LAYOUT_NDHWC, LAYOUT_NDHWC_HPS = 0, 1, 2
LAYOUT_BF16_NDHWC, LAYOUT_BF16_NDHWC_HPS, LAYOUT_NCDHW_RBF16 = 3, 4, 5
LAYOUT_SIGN4 = 6   # (B,D,W,H,C/4) uint8 LeakyReLU sign mask of a block output (see the header)
DRR_HU_INPUT, DRR_FLIP_W = 1, 2
WARP_USING_SCALE, WARP_BORDER, WARP_NEAREST = 1, 2, 4
NCC_CONFIGURED, NCC_SQUARED = 0, 1
MAX_VIEWS = 32

_p = C.c_void_p
_i, _i64, _f = C.c_int, C.c_int64, C.c_float

# name -> (restype, argtypes); mirrors include/liftreg_hip.h one to one
SIGNATURES = {
    "lr_strerror": (C.c_char_p, [_i]),
    "lr_abi_version": (_i, []),
    "lr_target_arch": (C.c_char_p, []),
    "lr_reload_switches": (_i, []),
    "lr_switch_name": (C.c_char_p, [_i]),
    "lr_drr_forward_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "lr_drr_forward_batch_f32": (_i, [_p, _i64, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "lr_hu_to_mu_f32": (_i, [_p, _p, _i64, _p]),
    "lr_drr_sample_coords_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "lr_backproject_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i64, _p]),
    "lr_backproject_coords_f32": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "lr_backproject_coords_poseless_f64": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "lr_sample_points_f64": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "lr_conv3d_packed_floats": (_i64, [_i, _i, _i]),
    "lr_conv3d_pack_weights_f32": (_i, [_p, _p, _i, _i, _i, _p]),
    "lr_conv3d_k3_lrelu_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _p]),
    "lr_conv3d_k3_lrelu_zphase_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _p]),
    "lr_conv3d_k3_lrelu_obs_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _i64, _p]),
    "lr_conv3d_k3_lrelu_mask_f32": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _p]),
    "lr_conv3d_dgrad_wgrad0_partial_floats": (_i64, [_i]),
    "lr_conv3d_dgrad_wgrad0_f32": (_i, [_p, _p, _p, _f, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "lr_conv3d_dgrad_wgrad0_split_f32": (_i, [_p, _p, _p, _f, _p, _i64, _p, _i64, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "lr_linear_lrelu_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _f, _p]),
    "lr_pca_reconstruct_f32": (_i, [_p, _p, _p, _p, _i, _i, _i64, _i64, _i64, _p]),
    "lr_pca_reconstruct_bf16basis_f32": (_i, [_p, _p, _p, _p, _i, _i, _i64, _i64, _i64, _p]),
    "lr_pca_bwd_coef_bf16basis_f32": (_i, [_p, _p, _p, _p, _i, _i, _i64, _i64, _i64, _i, _p]),
    "lr_conv3d_first_split_f32": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _p]),
    "lr_conv3d_first_split_obs_f32": (_i, [_p, _i64, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _i64, _p]),
    "lr_conv3d_pair01_packed_floats": (_i64, [_i, _i, _i]),
    "lr_conv3d_pair01_pack_f32": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "lr_conv3d_pair01_f32": (_i, [_p, _i64, _p, _i64, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _f, _i64, _p]),
    "lr_conv3d_pair01_train_f32": (_i, [_p, _i64, _p, _i64, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _f, _p]),
    "lr_conv3d_pair01_slab_f32": (_i, [_p, _i64, _p, _i64, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _f, _i64, _i, _i, _i, _i, _p]),
    "lr_pca_warp_f32": (_i, [_p] * 10 + [_i, _i, _i, _i, _i, _i, _i64, _i, _p]),
    "lr_pca_warp_bf16basis_f32": (_i, [_p] * 10 + [_i, _i, _i, _i, _i, _i, _i64, _i, _p]),
    "lr_pca_warp_slab_f32": (_i, [_p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i64, _i64, _i,
                                  _p, _p, _p, _p]),
    "lr_warp_trilinear_f32": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "lr_mask_compose_f32": (_i, [_p, _p, _p, _i64, _p]),
    "lr_ncc_moments_f32": (_i, [_p, _p, _p, _p, _i, _i64, _i, _p]),
    "lr_ncc_loss_from_moments": (_i, [_p, _p, _p, _i, _i64, _i, _i, _p]),
    "lr_disp_reg_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "lr_subspace_reg_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _p]),
    "lr_ncc_bwd_f32": (_i, [_p, _p, _p, _p, _p, _i, _i64, _i64, _i, _p]),
    "lr_warp_bwd_disp_f32": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "lr_warp_bwd_disp_acc_f32": (_i, [_p] * 9 + [_i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "lr_ncc_bwd_moments": (_i, [_p, _p, _p, _i, _i64, _i, _p]),
    "lr_warp_bwd_disp_ncc_f32": (_i, [_p] * 10 + [_i, _i, _i, _i, _i, _i, _i, _p]),
    "lr_pca_bwd_coef_f32": (_i, [_p, _p, _p, _p, _i, _i, _i64, _i64, _i64, _i, _p]),
    "lr_linear_bwd_f32": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _p]),
    "lr_lrelu_bwd_f32": (_i, [_p, _i, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _p]),
    "lr_conv3d_dgrad_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _i, _f, _p]),
    "lr_disp_reg_bwd_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "lr_conv3d_packed_bf16_bytes": (_i64, [_i, _i]),
    "lr_conv3d_pack_weights_bf16": (_i, [_p, _p, _i, _i, _p]),
    "lr_conv3d_k3_lrelu_bf16": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _p]),
    "lr_conv3d_k3_lrelu_obs_bf16": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i64, _p]),
    "lr_cast_f32_to_bf16": (_i, [_p, _p, _i64, _p]),
    "lr_conv3d_packed_bf16_planar_bytes": (_i64, [_i, _i]),
    "lr_conv3d_pack_weights_bf16_planar": (_i, [_p, _p, _i, _i, _p]),
    "lr_conv3d_first_bf16": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _p]),
    "lr_backproject_encin_bf16": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i64, _p]),
    "lr_conv3d_first_clin_bf16": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _i64, _p]),
    "lr_conv3d_first_mask_bf16": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _p]),
    "lr_conv3d_first_obs_bf16": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _i64, _p]),
    "lr_conv3d_dgrad_bf16": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _i, _f, _p]),
    "lr_conv3d_wgrad_bf16g_f32": (_i, [_p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "lr_normalize_clip_f32": (_i, [_p, _p, _i64, _f, _f, _p]),
    "lr_label_overlap_f32": (_i, [_p, _p, _f, _i64, _p, _i, _p, _p]),
    "lr_jacobi_det_stats_f32": (_i, [_p, _i, _i, _i, _i, _f, _f, _f, _p, _i, _p, _p]),
    "lr_conv3d_wgrad_partial_floats": (_i64, [_i, _i, _i, _i]),
    "lr_conv3d_wgrad_f32": (_i, [_p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
}


# the `#ifdef LR_EXPERIMENTAL` section of the header: bound only when the loaded library exports them (`make exp`)
EXPERIMENTAL_SIGNATURES = {
    "lr_conv3d_first_fused_bp_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _p]),
}


class LiftRegHipError(RuntimeError):
    pass


def build_library(force=False, quiet=True):
    """Compile libliftreg_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j8"] + (["-B"] if force else [])
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL if quiet else None)
    return LIB_PATH


_lib = None


def lib():
    """The loaded library; raises LiftRegHipError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LiftRegHipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C liftreg_amd/csrc` — liftreg_amd has no CPU/PyTorch fallback path")
        # PyTorch-ROCm wheels bundle their own libamdhip64.so.7; the device pointers and streams we are
        # handed live in THAT runtime.  Import torch first so the dynamic loader resolves our
        # DT_NEEDED libamdhip64.so.7 to the copy torch already loaded — loading ours first would put a
        # second HIP runtime (/opt/rocm) in the process, whose launches fail with hipErrorNoDevice.
        import torch  # noqa: F401
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        for name, (res, args) in EXPERIMENTAL_SIGNATURES.items():
            if hasattr(handle, name):
                fn = getattr(handle, name)
                fn.restype, fn.argtypes = res, args
        if handle.lr_target_arch() != b"gfx950":
            raise LiftRegHipError("libliftreg_hip.so was not built for gfx950")
        _lib = handle
    return _lib


def has_experimental():
    """True when the loaded library is the experimental build (include/liftreg_hip.h, last section)."""
    return all(hasattr(lib(), n) for n in EXPERIMENTAL_SIGNATURES)


def reload_switches():
    """Re-read the library's LIFTREG_* environment switches (it reads them once per process; include/liftreg_hip.h lists
    them): call after changing os.environ between two launches of one process (tests, A/B tools)."""
    return lib().lr_reload_switches()


# Tests and A/B tools flip switches between two launches of ONE process.  With AUTOSYNC on (tests/conftest.py sets it; never in
# production) every launch first compares the switches' environment values with the last snapshot and reloads on a change.
def _env_int(name, dflt=0):
    try:
        return int(os.environ.get(name, dflt))
    except ValueError:
        return dflt


AUTOSYNC = _env_int("LIFTREG_SWITCH_AUTOSYNC") != 0      # "0" and unset are off
_switch_names = None
_switch_snapshot = None


def sync_switches():
    global _switch_names, _switch_snapshot
    if _switch_names is None:
        h = lib()
        _switch_names = [h.lr_switch_name(i).decode() for i in range(h.lr_reload_switches())]
    snap = tuple(os.environ.get(n) for n in _switch_names)
    if snap != _switch_snapshot:
        lib().lr_reload_switches()
        _switch_snapshot = snap


def check(code, what):
    if code != LR_OK:
        raise LiftRegHipError(f"{what}: {lib().lr_strerror(code).decode()} ({code})")
