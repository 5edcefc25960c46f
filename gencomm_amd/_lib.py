"""ctypes binding of ``libgencomm_hip.so`` (C ABI in ``include/gencomm_hip.h``).

The product path has no fallback: if the shared library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os
import shutil
import subprocess
import threading
from typing import List, Tuple

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
CSRC_DIR = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.environ.get("GENCOMM_HIP_LIB", os.path.join(PKG_DIR, "libgencomm_hip.so"))  # override: diagnostic builds only
ABI_VERSION = 11
MODE_ARITH, MODE_SAMPLER, MODE_TILE_WANT, MODE_ENH_FUSE, MODE_CONV8H_MASK, MODE_XCD_REMAP, MODE_DATAFLOW, MODE_RESFUSE_EMU, MODE_TILE8, MODE_BWD_STREAMS, MODE_PERSIST = range(11)

_lock = threading.Lock()
_lib = None

c_float_p = C.c_void_p  # device pointers travel as plain integers
_i, _ll, _p = C.c_int, C.c_longlong, C.c_void_p

_SIGNATURES = {
    "gencomm_abi_version": (_i, []),
    "gencomm_last_error": (C.c_char_p, []),
    "gencomm_build_info": (C.c_char_p, []),
    "gencomm_set_mode": (_i, [_i, _ll]),
    "gencomm_get_mode": (_ll, [_i]),
    "gencomm_timer_num_kernels": (_i, []),
    "gencomm_timer_kernel_name": (C.c_char_p, [_i]),
    "gencomm_timer_start": (_i, [_i, _i]),
    "gencomm_timer_stop": (_i, [C.POINTER(C.c_double), C.POINTER(_i)]),
    "gencomm_timer_start_mask": (_i, [C.c_ulonglong, _i]),
    "gencomm_timer_stop_families": (_i, [C.POINTER(C.c_double), C.POINTER(_i), C.POINTER(C.c_double), _i]),
    "gencomm_klog_start": (_i, []),
    "gencomm_klog_stop": (_i, [C.c_char_p, _i]),
    "gencomm_clock_probe": (_i, [_p, _i, _p]),
    "gencomm_unet_num_params": (_i, [_i, _i, _i, _i]),
    "gencomm_unet_param_info": (_i, [_i, _i, _i, _i, _i, C.c_char_p, _i, C.POINTER(_ll), C.POINTER(_ll)]),
    "gencomm_unet_raw_floats": (_ll, [_i, _i, _i, _i]),
    "gencomm_unet_prepared_floats": (_ll, [_i, _i, _i, _i, _i]),
    "gencomm_unet_prepare": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "gencomm_unet_bwd_workspace_bytes": (_ll, [_i, _i, _i, _i, _i, _i, _i]),
    "gencomm_unet_fwd_train": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _ll, _p]),
    "gencomm_unet_fwd_train_step": (_i, [_p, _p, _p, _p, _i, _p, _p, C.c_ulonglong, _i, _i, _i, _i, _i, _i, _i, _i, _p, _ll, _p]),
    "gencomm_unet_bwd": (_i, [_p, _p, _p, _p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _ll, _p]),
    "gencomm_unet_bwd_chain": (_i, [_p, _p, _p, _p, _i, _p, C.c_float, C.c_float, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _ll, _p]),
    "gencomm_conv8_fwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "gencomm_denoise_workspace_bytes": (_ll, [_i, _i, _i, _i, _i, _i, _i]),
    "gencomm_dataflow_error": (_i, [_p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "gencomm_dataflow_words": (_i, [_p, _i, _i, _i, _i, _i, _i, _i, C.POINTER(C.c_uint), _i, _p]),
    "gencomm_unet_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _ll, _p]),
    "gencomm_denoise_fwd": (_i, [_p, _p, _p, _i, _p, _p, _p, _p, _p, C.c_ulonglong,
                                 _i, _i, _i, _i, _i, _i, _i, _i, _p, _ll, _p]),
    "gencomm_denoise_fwd_dseed": (_i, [_p, _p, _p, _i, _p, _p, _p, _p, _p, C.c_ulonglong, _p,
                                       _i, _i, _i, _i, _i, _i, _i, _i, _p, _ll, _p]),
    "gencomm_q_sample_fwd": (_i, [_p, _p, _i, _p, _p, C.c_ulonglong, C.c_uint, _p, _i, _i, _i, _i, _p]),
    "gencomm_step_noise_fwd": (_i, [_p, C.c_ulonglong, C.c_uint, _p, _i, _i, _i, _i, _i, _p]),
    "gencomm_enhancer_num_params": (_i, [_i]),
    "gencomm_enhancer_param_info": (_i, [_i, _i, C.c_char_p, _i, C.POINTER(_ll), C.POINTER(_ll)]),
    "gencomm_enhancer_raw_floats": (_ll, [_i]),
    "gencomm_enhancer_workspace_bytes": (_ll, [_i, _i, _i, _i]),
    "gencomm_enhancer_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _ll, _p]),
    "gencomm_msgext_num_params": (_i, [_i]),
    "gencomm_msgext_param_info": (_i, [_i, _i, C.c_char_p, _i, C.POINTER(_ll), C.POINTER(_ll)]),
    "gencomm_msgext_raw_floats": (_ll, [_i]),
    "gencomm_msgext_workspace_bytes": (_ll, [_i, _i, _i, _i]),
    "gencomm_msgext_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _ll, _p]),
    "gencomm_pillar_encode_fwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i,
                                       C.POINTER(C.c_float), C.POINTER(C.c_float), _p]),
    "gencomm_conv2d_prepare": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "gencomm_conv2d_prepared_floats": (_ll, [_i, _i, _i, _i, _i]),
    "gencomm_conv2d_fold": (_i, [_p, _p, _p, _p, _p, C.c_float, _i, _p, _p, _p]),
    "gencomm_conv2d_fwd": (_i, [_p, _p, _p, _p, _p] + [_i] * 13 + [_p]),
    "gencomm_conv2d_act_res_fwd": (_i, [_p, _p, _p, _p, _p, _p] + [_i] * 10 + [_p]),
    "gencomm_split3_attn_fwd": (_i, [_p] * 10 + [_i, _i, _i, _p]),
    "gencomm_warp_affine_bwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "gencomm_hgt_attn_bwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "gencomm_win_attn_bwd_scratch_floats": (_ll, [_i, _i, _i, _i, _i]),
    "gencomm_win_attn_bwd": (_i, [_p] * 7 + [_i] * 6 + [_p]),
    "gencomm_conv2d_wgrad": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "gencomm_conv2d_wgrad_scratch_floats": (_ll, [_i] * 8),
    "gencomm_conv2d_wgrad_ws": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _ll, _p]),
    "gencomm_gn_nchw_fwd": (_i, [_p, _p, _p, _p, _p, C.c_float, _i, _i, _i, _i, _i, _p]),
    "gencomm_ln_nchw_fwd": (_i, [_p, _p, _p, _p, C.c_float, _i, _i, _i, _i, _p]),
    "gencomm_ln_nchw_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, C.c_float, _i, _i, _i, _i, _p]),
    "gencomm_dwconv3x3_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "gencomm_dwconv3x3_wgrad": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "gencomm_conv3x3_c16_scratch_floats": (_ll, []),
    "gencomm_conv3x3_c16_fwd": (_i, [_p, _i, _p, _i, _p, _i, _p, _i, _i, _i, _p]),
    "gencomm_dwconv3x3_act_fwd": (_i, [_p, _i, _i, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "gencomm_dwconv3x3_act_wgrad": (_i, [_p, _i, _i, _p, _p, _p, _i, _i, _i, _i, _p]),
    "gencomm_gelu_bwd": (_i, [_p, _p, _p, _ll, _p]),
    "gencomm_lincomb_fwd": (_i, [_p, _p, _p, _p, C.c_float, C.c_float, C.c_float, _ll, _p]),
    "gencomm_ew_slice_fwd": (_i, [_i, _p, _p, _p, _p, _p, _p] + [_i] * 9 + [_p]),
    "gencomm_nc_scale_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _p]),
    "gencomm_nc_dot_fwd": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "gencomm_det_workspace_bytes": (C.c_longlong, [_i, _i, _i]),
    "gencomm_nms_workspace_bytes": (C.c_longlong, []),
    "gencomm_nms_max_candidates": (_i, []),
    "gencomm_det_decode_fwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, C.c_float, C.c_float, _i, _p, _p, _p, _p, _i, _p, C.c_longlong, _p]),
    "gencomm_nms_rotated_fwd": (_i, [_p, _p, _p, C.c_float, _i, _p, _p, _p, _p, _p, _p, C.c_longlong, _p]),
    "gencomm_bbox_overlaps_fwd": (_i, [_p, _p, _p, _i, _i, _p]),
    "gencomm_warp_affine_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "gencomm_hgt_attn_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "gencomm_win_attn_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "gencomm_iou3d_pairwise_fwd": (_i, [_p, _i, _p, _i, _i, _p, _p]),
    "gencomm_iou3d_max_boxes": (_i, []),
    "gencomm_iou3d_nms_workspace_bytes": (_ll, [_i]),
    "gencomm_iou3d_nms_fwd": (_i, [_p, _i, C.c_float, _i, _p, _p, _p, _ll, _p]),
    "gencomm_bn2d_train_fwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, C.c_float, C.c_float, _i, _i, _i, _i, _p, _p]),
    "gencomm_bn2d_train_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "gencomm_slot_max_fwd": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "gencomm_convbn_train_fwd": (_i, [_p] * 10 + [C.c_float, C.c_float, _i] + [_p] * 5 + [_i] * 8 + [_p]),
    "gencomm_convbn_train_bwd": (_i, [_p] * 9 + [_i] + [_p] * 9 + [_ll] + [_i] * 7 + [_p]),
    "gencomm_pfn_train_fwd": (_i, [_p, _p, _p, _p, _p, _p, _p, C.c_float, C.c_float, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "gencomm_pfn_train_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "gencomm_pfn_moment_doubles": (_ll, [_i]),
    "gencomm_pfn_bwd_scratch_doubles": (_ll, [_i, _i]),
    "gencomm_slot_max_bwd": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "gencomm_head_loss": (_i, [_p] * 10 + [_i] * 5 + [_p, C.c_double] + [C.c_float] * 7 + [_i, _p]),
    "gencomm_dcn_sample_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "gencomm_dcn_scatter_bwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "gencomm_dcn_scatter_scratch_floats": (_ll, [_i, _i, _i, _i]),
    "gencomm_dcn_scatter_bwd_ws": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _ll, _p]),
    "gencomm_sp_out_dims": (_i, [_p, _p, _p, _p, _p]),
    "gencomm_sp_index_workspace_bytes": (_ll, [_i]),
    "gencomm_sp_index_fwd": (_i, [_p, _i, _i, _p, _p, _p, _p, _ll, _p]),
    "gencomm_sp_sites_capacity": (_ll, [_i, _p, _p]),
    "gencomm_sp_sites_workspace_bytes": (_ll, [_i, _p, _p]),
    "gencomm_sp_sites_fwd": (_i, [_p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _ll, _p]),
    "gencomm_sp_rules_fwd": (_i, [_p, _i, _p, _i, _i, _p, _p, _p, _p, _p, _p]),
    "gencomm_sp_prepared_floats": (_ll, [_i, _i, _i]),
    "gencomm_sp_prepare": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "gencomm_sp_conv_fwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "gencomm_sp_dense_fwd": (_i, [_p, _p, _i, _i, _i, _p, _p, _p]),
    "gencomm_mean_vfe_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _p]),
    "gencomm_bnrow_train_fwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, C.c_float, C.c_float, _i, _i, _i, _p]),
    "gencomm_bnrow_train_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "gencomm_sp_rules_inv_fwd": (_i, [_p, _i, _p, _i, _i, _p, _p, _p, _p, _p, _p]),
    "gencomm_sp_wgrad": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "gencomm_voxelize_workspace_bytes": (_ll, [_i]),
    "gencomm_voxelize_fwd": (_i, [_p, _i, _i, C.POINTER(C.c_float), C.POINTER(C.c_float), _i, _i, _p, _p, _p, _p, _p, _ll, _p]),
    "gencomm_warp_attfuse_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "gencomm_warp_attfuse_bwd_scratch_floats": (_ll, [_i, _i, _i]),
    "gencomm_warp_attfuse_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "gencomm_warp_maxfuse_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "gencomm_warp_attfuse_tok_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)


class GenCommHipError(RuntimeError):
    pass


def hip_sources() -> List[str]:
    return [os.path.join(CSRC_DIR, "gencomm_abi.hip"), os.path.join(CSRC_DIR, "gencomm_abi_aux.hip")]


def source_hash() -> str:
    """16 hex digits over the HIP sources and the C header: compiled into the library (`gencomm_build_info()` ends in
    ` src=<hash>`) so that committed measurements (profiles/*_pmc_traffic.json) can say which kernels they were taken on."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC_DIR)) + [os.path.join(REPO_DIR, "include", "gencomm_hip.h")]:
        path = f if os.path.isabs(f) else os.path.join(CSRC_DIR, f)
        if os.path.isfile(path):
            h.update(os.path.basename(path).encode())
            h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def library_src_hash() -> str:
    """The `src=` stamp of the LOADED library ("" for a library built before the stamp existed)."""
    info = lib().gencomm_build_info().decode()
    return info.rsplit(" src=", 1)[1] if " src=" in info else ""


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP library in-tree for gfx950 (cross-compiles without a GPU)."""
    srcs = hip_sources()
    deps = [os.path.join(CSRC_DIR, f) for f in os.listdir(CSRC_DIR)] + [os.path.join(REPO_DIR, "include", "gencomm_hip.h"), os.path.abspath(__file__)]
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps):
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    # -packed-fp32-ops: no v_pk_{fma,mul,add}_f32. With them, conv8h_kernel's epilogue `acc * inv_scale + bias` compiled to
    # `v_pk_fma_f32 v[0:1], v[0:1], s[22:23], v[22:23] op_sel:[0,0,1]` and sporadically lost the bias in the low half on
    # lanes 48..63 whenever two workgroups shared a SIMD (never with one workgroup per CU); the same source built
    # without packed ops is exact in every run (tools/conv8_unit.py), and the hot kernels are not slower for it.
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall",
             "-Wno-unused-function", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
    # diagnostic builds (A/B variants under tools/): GENCOMM_HIP_LIB names the output, GENCOMM_EXTRA_FLAGS adds -D switches; such a
    # library carries its switches in gencomm_build_info() and a different source stamp is not needed (the flags differ)
    extra = os.environ.get("GENCOMM_EXTRA_FLAGS", "").split()
    flags += extra
    define = '-DGENCOMM_BUILD_FLAGS="' + " ".join(flags) + " src=" + source_hash() + '"'
    obj_dir = os.path.join(PKG_DIR, "_build") if LIB_PATH == os.path.join(PKG_DIR, "libgencomm_hip.so") else LIB_PATH + ".obj"
    os.makedirs(obj_dir, exist_ok=True)
    # one object per translation unit, compiled concurrently (the hot path's unit takes ~2 min, the rocPRIM one ~1 min)
    procs, objs = [], []
    for src in srcs:
        obj = os.path.join(obj_dir, os.path.splitext(os.path.basename(src))[0] + ".o")
        objs.append(obj)
        cmd = [hipcc, *flags, define, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
    failed = [(cmd, pr.returncode) for cmd, pr in procs if pr.wait() != 0]   # every compile has finished before anything is raised
    if failed:
        raise subprocess.CalledProcessError(failed[0][1], failed[0][0])
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    global _lib
    _lib = None
    return LIB_PATH


def lib() -> C.CDLL:
    """The loaded library; raises GenCommHipError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise GenCommHipError(
                    f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(hipcc --offload-arch=gfx950). There is no CPU fallback for the GenComm hot path.")
            l = C.CDLL(LIB_PATH)
            for name, (res, args) in _SIGNATURES.items():
                fn = getattr(l, name)  # AttributeError here = ABI mismatch, fail loudly
                fn.restype, fn.argtypes = res, args
            if l.gencomm_abi_version() != ABI_VERSION:
                raise GenCommHipError(f"ABI version mismatch: library {l.gencomm_abi_version()}, binding {ABI_VERSION}")
            if b"-packed-fp32-ops" not in l.gencomm_build_info():
                raise GenCommHipError(f"{LIB_PATH} was not built with '-Xclang -target-feature -Xclang -packed-fp32-ops' "
                                      f"(build info: {l.gencomm_build_info().decode()!r}); rebuild with gencomm_amd._lib.build(force=True)")
            _lib = l
    return _lib


class mode:
    """``with _lib.mode(_lib.MODE_ARITH, 1): ...`` -- set a library mode (include/gencomm_hip.h) for a block, restoring the
    previous value afterwards. Modes are process-wide: do not flip them while another thread is inside the library."""

    def __init__(self, key: int, value: int):
        self.key, self.value = key, int(value)

    def __enter__(self):
        l = lib()
        self.prev = l.gencomm_get_mode(self.key)
        check(l.gencomm_set_mode(self.key, self.value), "gencomm_set_mode")
        return self

    def __exit__(self, *exc):
        check(lib().gencomm_set_mode(self.key, self.prev), "gencomm_set_mode")
        return False


class kernel_log:
    """``with _lib.kernel_log() as kl: ...; kl.counts`` -- the kernel instantiations the hot path's launch sites chose inside the
    block (name -> number of launches). Diagnostic, process-wide like the modes."""

    def __enter__(self):
        check(lib().gencomm_klog_start(), "gencomm_klog_start")
        self.counts = {}
        return self

    def __exit__(self, *exc):
        buf = C.create_string_buffer(1 << 16)
        check(lib().gencomm_klog_stop(buf, len(buf)), "gencomm_klog_stop")
        for line in buf.value.decode().splitlines():
            name, _, cnt = line.rpartition("\t")
            self.counts[name] = int(cnt)
        return False


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise GenCommHipError(f"{what} failed (status {rc}): {lib().gencomm_last_error().decode()}")


def check_size(v: int, what: str) -> int:
    if v < 0:
        raise GenCommHipError(f"{what} failed: {lib().gencomm_last_error().decode()}")
    return int(v)


def unet_param_table(Cch: int, levels: int, res_blocks: int, attn_mask: int = 0) -> List[Tuple[str, int, int]]:
    l = lib()
    n = check_size(l.gencomm_unet_num_params(Cch, levels, res_blocks, attn_mask), "gencomm_unet_num_params")
    out, buf, numel, off = [], C.create_string_buffer(128), _ll(), _ll()
    for i in range(n):
        check(l.gencomm_unet_param_info(Cch, levels, res_blocks, attn_mask, i, buf, 128, C.byref(numel), C.byref(off)), "gencomm_unet_param_info")
        out.append((buf.value.decode(), int(numel.value), int(off.value)))
    return out


def enhancer_param_table(Cch: int) -> List[Tuple[str, int, int]]:
    l = lib()
    n = check_size(l.gencomm_enhancer_num_params(Cch), "gencomm_enhancer_num_params")
    out, buf, numel, off = [], C.create_string_buffer(128), _ll(), _ll()
    for i in range(n):
        check(l.gencomm_enhancer_param_info(Cch, i, buf, 128, C.byref(numel), C.byref(off)), "gencomm_enhancer_param_info")
        out.append((buf.value.decode(), int(numel.value), int(off.value)))
    return out


def msgext_param_table(Cch: int) -> List[Tuple[str, int, int]]:
    l = lib()
    n = check_size(l.gencomm_msgext_num_params(Cch), "gencomm_msgext_num_params")
    out, buf, numel, off = [], C.create_string_buffer(128), _ll(), _ll()
    for i in range(n):
        check(l.gencomm_msgext_param_info(Cch, i, buf, 128, C.byref(numel), C.byref(off)), "gencomm_msgext_param_info")
        out.append((buf.value.decode(), int(numel.value), int(off.value)))
    return out
