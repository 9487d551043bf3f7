"""ctypes binding of ``libdpvo_hip.so`` (the C ABI declared in ``include/dpvo_hip.h``).

The library is the product: there is NO fallback.  If the shared object is missing or a symbol is
absent, importing/using any op raises immediately (``DPVOHipError``), on CPU-only hosts as well as
on the GPU box.  Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C dpvo_amd/csrc``.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DPVO_HIP_LIB") or os.path.join(_HERE, "libdpvo_hip.so")      # (override: development builds)

F16, F32 = 0, 1
ABI_VERSION = 8         # == DPVO_ABI_VERSION of include/dpvo_hip.h this binding (struct layouts, signatures) was written against

# every symbol include/dpvo_hip.h declares (tests/test_capi.py checks the .so exports all of them)
SYMBOLS = [
    "dpvo_abi_version",
    "dpvo_corr_forward", "dpvo_corr_pyramid_forward", "dpvo_patchify_forward", "dpvo_patchify_bilinear",
    "dpvo_reproject", "dpvo_flow_mag", "dpvo_loop_flow", "dpvo_motionmag", "dpvo_motionmag_status", "dpvo_point_cloud", "dpvo_point_cloud_motionmag",
    "dpvo_normalize_scratch_bytes", "dpvo_normalize",
    "dpvo_se3_inv", "dpvo_se3_mul", "dpvo_se3_act4", "dpvo_se3_exp", "dpvo_se3_log",
    "dpvo_plan_layout", "dpvo_plan_workspace_bytes", "dpvo_plan_build", "dpvo_plan_build_ranged", "dpvo_plan_build_window", "dpvo_plan_build_window_flow",
    "dpvo_plan_wide_workspace_bytes", "dpvo_plan_build_wide",
    "dpvo_neighbors_workspace_bytes", "dpvo_neighbors",
    "dpvo_softagg",
    "dpvo_update_fused_pack_bytes", "dpvo_update_fused_pack", "dpvo_update_fused_workspace_bytes", "dpvo_update_forward_fused", "dpvo_update_forward_fused_rows", "dpvo_update_fused_default_tiling",
    "dpvo_ba_workspace_bytes", "dpvo_ba",
    "dpvo_gba_workspace_bytes", "dpvo_gba_linearize", "dpvo_gba_relinearize", "dpvo_gba_retract", "dpvo_gba_solve_workspace_bytes", "dpvo_gba_solve",
    "dpvo_solve_system_workspace_bytes", "dpvo_solve_system",
    "dpvo_normalize_image", "dpvo_patch_colors", "dpvo_store_features", "dpvo_append_edges", "dpvo_gather_edges", "dpvo_gather_edges2",
    "dpvo_motion_model", "dpvo_median_depth", "dpvo_frame_patches", "dpvo_frame_state", "dpvo_frame_state_part", "dpvo_keyframe_step", "dpvo_frame_update", "dpvo_debug_stamp",
    "dpvo_encoders_workspace_bytes", "dpvo_encoders_forward", "dpvo_encoders_forward_hold", "dpvo_pool4_nhwc",
]


# libdpvo_hip_cmp.so (include/dpvo_hip_cmp.h): the COMPARATOR implementation of the update operator (launch by launch, update.hip) -- test and measurement
# partner of the product's seven-launch operator, never loaded by the tracker (cmp_lib() below)
CMP_LIB_PATH = os.environ.get("DPVO_HIP_CMP_LIB") or os.path.join(_HERE, "libdpvo_hip_cmp.so")
CMP_SYMBOLS = ["dpvo_linear", "dpvo_layernorm", "dpvo_gather_add", "dpvo_heads", "dpvo_heads_target",
               "dpvo_update_workspace_bytes", "dpvo_update_forward"]


class DPVOHipError(RuntimeError):
    pass


class FrameState(ctypes.Structure):
    """dpvo_frame_state_t"""
    _fields_ = ([(k, ctypes.c_void_p) for k in (
        "fmap", "imap", "img_u8", "coords", "xs", "ys", "depth", "intrinsics", "gmap_slot", "imap_slot", "patches_slot",
        "colors_slot", "intrinsics_slot", "index_row", "index_map", "poses", "patches_all", "fmap2_slot", "ii", "jj", "kk",
        "net", "ix")] +
        [(k, ctypes.c_int64) for k in ("frame_next", "m_next", "E0", "n_new")] +
        [(k, ctypes.c_float) for k in ("res", "mm_scale")] +
        [(k, ctypes.c_int32) for k in ("M", "h", "w", "H", "W", "CF", "CI", "P", "mm_n", "md_n", "ap_n", "ap_r", "D")])


class Ring(ctypes.Structure):
    """dpvo_ring_t"""
    _fields_ = [("base", ctypes.c_void_p), ("slot_bytes", ctypes.c_int64), ("ring", ctypes.c_int64)]


class KeyframeStep(ctypes.Structure):
    """dpvo_keyframe_step_t"""
    _fields_ = ([(k, ctypes.c_void_p) for k in (
        "ii", "jj", "kk", "net", "target", "weight", "ii_b", "jj_b", "kk_b", "net_b", "target_b", "weight_b",
        "ii_inac", "jj_inac", "kk_inac", "target_inac", "weight_inac")] +
        [("inac_room", ctypes.c_int64)] +
        [(k, ctypes.c_void_p) for k in ("flow4", "poses", "delta_pose", "keep_idx", "rem_idx", "keep_rows", "result", "result_host")] + [("host_words", ctypes.c_int32)] +
        [("ring", Ring * 8), ("n_ring", ctypes.c_int32), ("E", ctypes.c_int64)] +
        [(k, ctypes.c_int32) for k in ("n", "M", "D", "keyframe_index", "removal_window", "loop_closure",
                                       "optimization_window", "forced")] +
        [("keyframe_thresh", ctypes.c_float)])


class FrameUpdate(ctypes.Structure):
    """dpvo_frame_update_t"""
    _fields_ = ([("kf", KeyframeStep), ("fs", ctypes.c_void_p), ("ev_fs", ctypes.c_void_p), ("ev_enc", ctypes.c_void_p),
                 ("fmap_spec", ctypes.c_void_p), ("ev_record", ctypes.c_void_p),
                 ("ev_update_done", ctypes.c_void_p),
                 ("plan_stream", ctypes.c_void_p), ("ev_plan_fork", ctypes.c_void_p), ("ev_plan_done", ctypes.c_void_p),
                 ("fs_auto", ctypes.c_int32),
                 ("index_map", ctypes.c_void_p), ("net", ctypes.c_void_p), ("net_rows", ctypes.c_void_p), ("n_kept", ctypes.c_int64)] +
                [(k, ctypes.c_void_p) for k in ("poses", "patches", "intrinsics", "points", "ix", "gmap", "fmap1", "fmap2",
                                                "imap", "upd", "coords", "corr", "delta", "plan", "ws_plan", "ws_update", "ws_ba")] +
                [(k, ctypes.c_size_t) for k in ("ws_plan_bytes", "ws_update_bytes", "ws_ba_bytes")] +
                [("result_dev", ctypes.c_void_p), ("ev", ctypes.c_void_p * 4), ("m", ctypes.c_int64), ("n_buffer", ctypes.c_int64)] +
                [(k, ctypes.c_int32) for k in ("P", "pmem", "mem", "H0", "W0", "H1", "W1", "patch_lifetime", "ba_window",
                                               "iterations")] +
                [("lmbda", ctypes.c_float), ("mm_beta", ctypes.c_float), ("loop_out", ctypes.c_void_p), ("ev_loop", ctypes.c_void_p),
                 ("loop_freq", ctypes.c_int32), ("loop_max_age", ctypes.c_int32)])


class PlanLayout(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int64) for n in
                ("perm_k", "ku", "kx", "patch_off", "ix", "jx", "perm_p", "pu", "pair_off", "pair_ij", "counts",
                 "total_ints", "flow")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DPVOHipError(
                f"{LIB_PATH} not found: the HIP extension is required (no CPU/torch fallback exists). "
                "Build it with `make -C dpvo_amd/csrc` (hipcc --offload-arch=gfx950).")
        L = ctypes.CDLL(LIB_PATH)
        for s in SYMBOLS:
            if not hasattr(L, s):
                raise DPVOHipError(f"libdpvo_hip.so does not export {s}")
        for s in ("dpvo_plan_workspace_bytes", "dpvo_plan_wide_workspace_bytes", "dpvo_normalize_scratch_bytes", "dpvo_neighbors_workspace_bytes", "dpvo_ba_workspace_bytes",
                  "dpvo_gba_workspace_bytes", "dpvo_gba_solve_workspace_bytes", "dpvo_solve_system_workspace_bytes", "dpvo_encoders_workspace_bytes",
                  "dpvo_update_fused_workspace_bytes", "dpvo_update_fused_pack_bytes"):
            getattr(L, s).restype = ctypes.c_size_t
        L.dpvo_abi_version.restype = ctypes.c_int
        if L.dpvo_abi_version() != ABI_VERSION:
            raise DPVOHipError(f"{LIB_PATH} has ABI version {L.dpvo_abi_version()}, this binding needs {ABI_VERSION}: rebuild it "
                               "(`make -C dpvo_amd/csrc`)")
        _lib = L
    return _lib


_cmp = None


def cmp_lib():
    """the comparator library (launch-by-launch and patch-major update operators); loaded on first use, by tests / tools only"""
    global _cmp
    if _cmp is None:
        lib()                                    # (the comparator links against the product library)
        if not os.path.exists(CMP_LIB_PATH):
            raise DPVOHipError(f"{CMP_LIB_PATH} not found: build it with `make -C dpvo_amd/csrc`")
        C = ctypes.CDLL(CMP_LIB_PATH)
        for s in CMP_SYMBOLS:
            if not hasattr(C, s):
                raise DPVOHipError(f"libdpvo_hip_cmp.so does not export {s}")
        for s in ("dpvo_update_workspace_bytes",):
            getattr(C, s).restype = ctypes.c_size_t
        _cmp = C
    return _cmp


def check(rc, what):
    if rc != 0:
        kind = {-1: "invalid argument", -2: "unsupported configuration", -3: "workspace too small"}.get(
            rc, f"hipError_t {rc}")
        raise DPVOHipError(f"{what} failed: {kind}")


def ptr(t):
    """device pointer of a tensor (or NULL)"""
    if t is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t.data_ptr())


def row_ptr(t, i):
    """device pointer of t[i] without creating the view tensor (hot host paths)"""
    return ctypes.c_void_p(t.data_ptr() + int(i) * t.stride(0) * t.element_size())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """the current HIP stream of the current device as a void* (torch.cuda.current_stream() costs ~9 us of Python per
    call, ~40 calls per frame; the raw accessor is a plain C call)"""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def i64(v):
    return ctypes.c_int64(int(v))


def i32(v):
    return ctypes.c_int(int(v))


def f32(v):
    return ctypes.c_float(float(v))


def strides(t, dims):
    arr = (ctypes.c_int64 * len(dims))(*[t.stride(d) for d in dims])
    return arr


def dtype_code(dt):
    if dt == torch.float16:
        return F16
    if dt == torch.float32:
        return F32
    raise DPVOHipError(f"unsupported feature dtype {dt} (float16 / float32 only)")


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise DPVOHipError("dpvo_amd ops run on the GPU only (tensor on %s); there is no CPU path" % t.device)


def plan_layout(E):
    L = PlanLayout()
    check(lib().dpvo_plan_layout(i64(E), ctypes.byref(L)), "dpvo_plan_layout")
    return L
