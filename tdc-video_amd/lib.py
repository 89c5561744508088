"""ctypes binding of libtdc_hip.so (C ABI in include/tdc_hip.h).  No CPU fallback: every entry point raises if the
library cannot be loaded, and compute calls raise on a non-zero return code."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libtdc_hip.so")

F16, BF16 = 0, 1
ACT_NONE, ACT_GELU_ERF, ACT_GELU_TANH, ACT_SWIGLU = 0, 1, 2, 3


class RowMap(C.Structure):
    _fields_ = [("seg", C.c_int), ("stride", C.c_int), ("off", C.c_int), ("inner", C.c_int)]


class GemmDesc(C.Structure):
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int), ("W", C.c_void_p), ("ldw", C.c_int),
                ("C", C.c_void_p), ("ldc", C.c_int), ("bias", C.c_void_p), ("res", C.c_void_p), ("ldres", C.c_int),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("dtype", C.c_int), ("out_f32", C.c_int), ("res_f32", C.c_int), ("act", C.c_int),
                ("a_map", RowMap), ("c_map", RowMap), ("r_map", RowMap),
                ("x16", C.c_void_p), ("ldx16", C.c_int), ("ln_part", C.c_void_p),
                ("ln_stats", C.c_void_p), ("ln_c1", C.c_void_p), ("in_fp8", C.c_int),
                ("out_fp8", C.c_int), ("out_stats", C.c_void_p), ("out_w2max", C.c_float), ("out_bmax", C.c_float),
                ("out_wscale", C.c_float), ("c_pad8", C.c_int), ("c16_dtype_p1", C.c_int)]


class LnDesc(C.Structure):
    _fields_ = [("x", C.c_void_p), ("ldx", C.c_int), ("x_f32", C.c_int),
                ("y16", C.c_void_p), ("ldy16", C.c_int), ("y32", C.c_void_p), ("ldy32", C.c_int),
                ("gamma", C.c_void_p), ("beta", C.c_void_p), ("eps", C.c_float),
                ("add", C.c_void_p), ("ldadd", C.c_int), ("add_period", C.c_int), ("add_mode", C.c_int),
                ("rows", C.c_int), ("cols", C.c_int), ("dtype", C.c_int),
                ("x_map", RowMap), ("y_map", RowMap),
                ("y8", C.c_void_p), ("ldy8", C.c_int), ("y8_stats", C.c_void_p), ("y8_wscale", C.c_float),
                ("x_dtype_p1", C.c_int)]


class AttnDesc(C.Structure):
    _fields_ = [("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("o", C.c_void_p),
                ("q_bs", C.c_longlong), ("k_bs", C.c_longlong), ("v_bs", C.c_longlong), ("o_bs", C.c_longlong),
                ("q_rs", C.c_int), ("k_rs", C.c_int), ("v_rs", C.c_int), ("o_rs", C.c_int),
                ("batch", C.c_int), ("heads", C.c_int), ("head_dim", C.c_int), ("sq", C.c_int), ("sk", C.c_int),
                ("scale", C.c_float), ("dtype", C.c_int),
                ("bias", C.c_void_p), ("bias_hs", C.c_longlong), ("bias_rs", C.c_int),
                ("gate", C.c_void_p), ("gate_rs", C.c_int),
                ("key_mask", C.c_void_p), ("key_mask_bs", C.c_longlong), ("form", C.c_int)]


class GatherTables(C.Structure):
    _fields_ = [("base", C.c_void_p * 4), ("ld", C.c_int * 4)]


class SvaAttnDesc(C.Structure):
    _fields_ = [("q", C.c_void_p), ("ldq", C.c_int), ("kv", C.c_void_p * 2), ("ldkv", C.c_int),
                ("mask", C.c_void_p), ("out", C.c_void_p), ("ldo", C.c_int),
                ("T", C.c_int), ("side", C.c_int), ("r", C.c_int), ("n_towers", C.c_int), ("dim", C.c_int),
                ("heads", C.c_int), ("dtype", C.c_int)]


class QEmbedDesc(C.Structure):
    _fields_ = [("query", C.c_void_p), ("ldq", C.c_int), ("qsrc", C.c_void_p),
                ("word", C.c_void_p), ("pos", C.c_void_p), ("ldw", C.c_int), ("ids", C.c_void_p), ("Lt", C.c_int),
                ("gamma", C.c_void_p), ("beta", C.c_void_p), ("eps", C.c_float),
                ("h32", C.c_void_p), ("h16", C.c_void_p), ("ld", C.c_int),
                ("F", C.c_int), ("K", C.c_int), ("cols", C.c_int), ("dtype", C.c_int)]


class Lin(C.Structure):
    _fields_ = [("w", C.c_void_p), ("b", C.c_void_p), ("n", C.c_int), ("k", C.c_int)]


class VitLayer(C.Structure):
    _fields_ = [("ln1_g", C.c_void_p), ("ln1_b", C.c_void_p), ("ln2_g", C.c_void_p), ("ln2_b", C.c_void_p),
                ("qkv", Lin), ("out", Lin), ("fc1", Lin), ("fc2", Lin),
                ("qkv_c1", C.c_void_p), ("fc1_c1", C.c_void_p),
                ("qkv_wscale", C.c_float), ("fc1_wscale", C.c_float), ("zeros", C.c_void_p),
                ("out_wscale", C.c_float), ("fc2_wscale", C.c_float),
                ("fc1_w2max", C.c_float), ("fc1_bmax", C.c_float)]


class VitModel(C.Structure):
    _fields_ = [("dtype", C.c_int), ("dim", C.c_int), ("heads", C.c_int), ("head_dim", C.c_int),
                ("n_layers", C.c_int), ("patch", C.c_int), ("has_cls", C.c_int), ("act", C.c_int),
                ("eps", C.c_float), ("patch_lin", Lin), ("pos", C.c_void_p), ("ldpos", C.c_int),
                ("cls_row", C.c_void_p), ("lnf_g", C.c_void_p), ("lnf_b", C.c_void_p),
                ("layers_host", C.POINTER(VitLayer)), ("fused", C.c_int), ("fp8", C.c_int), ("out_dtype_p1", C.c_int),
                ("res_dtype_p1", C.c_int)]


class AuxProj(C.Structure):
    _fields_ = [("fc1", Lin), ("fc2", Lin), ("ln_g", C.c_void_p), ("ln_b", C.c_void_p)]


class SvaLayer(C.Structure):
    _fields_ = [("proj_context", Lin), ("proj_in_c", Lin), ("proj_in_q", Lin),
                ("pos", C.c_void_p * 2), ("ldpos", C.c_int), ("kv", Lin * 2),
                ("q_ln_g", C.c_void_p), ("q_ln_b", C.c_void_p), ("q_proj", Lin), ("o_proj", Lin),
                ("norm_g", C.c_void_p), ("norm_b", C.c_void_p), ("out1", Lin), ("out2", Lin)]


class ConnectorModel(C.Structure):
    _fields_ = [("dtype", C.c_int), ("C", C.c_int), ("side", C.c_int), ("r", C.c_int), ("heads", C.c_int),
                ("n_layers", C.c_int), ("aux", AuxProj * 2), ("vision_query", C.c_void_p),
                ("ones_C", C.c_void_p), ("zeros_C", C.c_void_p), ("layers_host", C.POINTER(SvaLayer)),
                ("mm1", Lin), ("mm2", Lin)]


class QformerLayer(C.Structure):
    _fields_ = [("qkv", Lin), ("attn_out", Lin), ("attn_ln_g", C.c_void_p), ("attn_ln_b", C.c_void_p),
                ("has_cross", C.c_int), ("cross_idx", C.c_int),
                ("cross_q", Lin), ("cross_out", Lin), ("cross_ln_g", C.c_void_p), ("cross_ln_b", C.c_void_p),
                ("fq1", Lin), ("fq2", Lin), ("fq_ln_g", C.c_void_p), ("fq_ln_b", C.c_void_p),
                ("ft1", Lin), ("ft2", Lin), ("ft_ln_g", C.c_void_p), ("ft_ln_b", C.c_void_p),
                ("cross_q_tiled", C.c_void_p), ("cross_out_tiled", C.c_void_p)]


class QformerModel(C.Structure):
    _fields_ = [("dtype", C.c_int), ("dim", C.c_int), ("heads", C.c_int), ("n_layers", C.c_int), ("H", C.c_int),
                ("eps", C.c_float), ("word", C.c_void_p), ("pos", C.c_void_p), ("ldw", C.c_int),
                ("emb_ln_g", C.c_void_p), ("emb_ln_b", C.c_void_p), ("cross_kv", Lin), ("vision_proj", Lin),
                ("layers_host", C.POINTER(QformerLayer)),
                ("cross_k", Lin), ("cross_v", Lin), ("cross_bv", C.c_void_p), ("xattn_mode", C.c_int)]


class XattnDesc(C.Structure):
    _fields_ = [("h16", C.c_void_p), ("h32", C.c_void_p), ("ldh", C.c_int),
                ("F", C.c_int), ("K", C.c_int), ("S", C.c_int),
                ("wq", C.c_void_p), ("bq", C.c_void_p), ("wo", C.c_void_p), ("bo", C.c_void_p),
                ("k", C.c_void_p), ("ldk", C.c_int),
                ("vt", C.c_void_p), ("ldvt", C.c_longlong), ("bv", C.c_void_p),
                ("Nenc", C.c_int), ("ln_g", C.c_void_p), ("ln_b", C.c_void_p), ("eps", C.c_float),
                ("dim", C.c_int), ("heads", C.c_int), ("scale", C.c_float), ("dtype", C.c_int),
                ("ctx", C.c_void_p), ("ldctx", C.c_int), ("res16", C.c_int)]


class ProfRec(C.Structure):
    _fields_ = [("kind", C.c_int), ("tag", C.c_int), ("ms", C.c_float), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("act", C.c_int), ("res", C.c_int), ("out_f32", C.c_int), ("W", C.c_void_p), ("flops", C.c_double)]


PROF_GEMM, PROF_ATTN, PROF_LN, PROF_XATTN = 1, 2, 3, 4
PROF_TAG_XATTN_BLOCK = 1

# name -> (restype, argtypes); every symbol include/tdc_hip.h declares
SIGNATURES = {
    "tdc_gemm": (C.c_int, [C.POINTER(GemmDesc), C.c_void_p]),
    "tdc_gemm_set_debug": (C.c_int, [C.c_int]),
    "tdc_gemm_set_persistent_grid": (C.c_int, [C.c_int]),
    "tdc_ln_finalize": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "tdc_quantize_rows_fp8": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                        C.c_float, C.c_void_p]),
    "tdc_layernorm": (C.c_int, [C.POINTER(LnDesc), C.c_void_p]),
    "tdc_attention": (C.c_int, [C.POINTER(AttnDesc), C.c_void_p]),
    "tdc_im2col": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                             C.c_void_p]),
    "tdc_set_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "tdc_set_rows16": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "tdc_resample_tokens": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "tdc_frame_cossim": (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "tdc_frame_cossim_scratch_floats": (C.c_size_t, [C.c_int]),
    "tdc_token_mean": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "tdc_adaptive_pool_tokens": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                           C.c_void_p, C.c_int, C.c_void_p]),
    "tdc_gather_rows": (C.c_int, [C.POINTER(GatherTables), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                  C.c_int, C.c_void_p]),
    "tdc_l2_normalize": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "tdc_sva_attention": (C.c_int, [C.POINTER(SvaAttnDesc), C.c_void_p]),
    "tdc_qformer_embed": (C.c_int, [C.POINTER(QEmbedDesc), C.c_void_p]),
    "tdc_vit_workspace_bytes": (C.c_size_t, [C.POINTER(VitModel), C.c_int, C.c_int, C.c_int]),
    "tdc_vit_fwd": (C.c_int, [C.POINTER(VitModel), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t,
                              C.c_void_p]),
    "tdc_fill_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "tdc_connector_workspace_bytes": (C.c_size_t, [C.POINTER(ConnectorModel), C.c_int]),
    "tdc_connector_fwd": (C.c_int, [C.POINTER(ConnectorModel), C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "tdc_qformer_workspace_bytes": (C.c_size_t, [C.POINTER(QformerModel), C.c_int, C.c_int, C.c_int, C.c_int]),
    "tdc_qformer_fwd": (C.c_int, [C.POINTER(QformerModel), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                  C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t,
                                  C.c_void_p]),
    "tdc_qformer_xattn_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "tdc_qformer_xattn": (C.c_int, [C.POINTER(XattnDesc), C.c_void_p]),
    "tdc_qformer_xattn_tile_weight": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "tdc_preprocess_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "tdc_preprocess_frames": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                        C.c_void_p]),
    "tdc_fbank_frames": (C.c_int, [C.c_longlong]),
    "tdc_fbank": (C.c_int, [C.c_void_p, C.c_int, C.c_longlong, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]),
    "tdc_relpos_gate": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "tdc_profile_start": (C.c_int, [C.c_int]),
    "tdc_profile_stop": (C.c_int, [C.POINTER(ProfRec), C.c_int]),
    "tdc_profile_tag": (C.c_int, [C.c_int]),
    "tdc_version": (C.c_char_p, []),
    "tdc_device_info": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_size_t)]),
}

_lib = None


class TdcHipError(RuntimeError):
    pass


def _build_once():
    """Fresh checkout: compile the library (hipcc, ~2.5 min) exactly once even when several ranks import at the same time
    (file lock; the losers find the finished .so).  Never rebuilds an existing library - staleness is build.py's business."""
    import fcntl
    import importlib.util
    try:
        lock = open(os.path.join(HERE, ".build.lock"), "w")
    except OSError as e:      # read-only install: nothing can be built here
        raise TdcHipError("libtdc_hip.so is missing and %s is not writable (%s): build it with `python -c 'import "
                          "__graft_entry__ as g; g.build()'` in a writable checkout" % (HERE, e))
    try:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not os.path.exists(LIB_PATH):
            spec = importlib.util.spec_from_file_location("tdc_build", os.path.join(HERE, "build.py"))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            mod.build()
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def load():
    """Load libtdc_hip.so (built in-tree by build.py; compiled here, once, when it is missing).  Raises if it cannot be
    loaded: there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: its bundled HIP runtime (libamdhip64) must be the one this process uses; loading ours first would
    # pull /opt/rocm's copy in and the two runtimes do not share devices/streams (hipErrorNoDevice at first launch).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        _build_once()
    if not os.path.exists(LIB_PATH):
        raise TdcHipError("libtdc_hip.so not found at %s and could not be built: run `python -c 'import __graft_entry__ "
                          "as g; g.build()'` (the product path has no CPU fallback)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise TdcHipError("%s failed with code %d" % (what, rc))
