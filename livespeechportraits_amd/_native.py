"""ctypes binding of liblspf2f.so (include/lspf2f.h, include/lspa2h.h, include/lsplle.h, include/lsprnn.h, include/lspraster.h, include/lspmel.h, include/lspunet.h).

There is deliberately no fallback: if the shared library is missing or does not
load, importing the hot path raises -- a GPU box must never silently run
something else (see DESIGN.md "No CPU path").
"""
from __future__ import annotations

import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t,
                    c_uint32, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblspf2f.so")
ABI_VERSION = 1

OK = 0
ERR_NAMES = {
    -1: "INVALID_ARGUMENT", -2: "UNSUPPORTED", -3: "MISSING_TENSOR", -4: "SHAPE",
    -5: "STATE", -6: "HIP", -7: "NO_DEVICE",
}
VARIANT_IDS = {"normal": 0, "large": 1}
DTYPE_IDS = {"f32": 0, "bf16": 1, "f16": 2}
FORM_IDS = {"rows": 0, "fullk": 1, "fullk2": 2, "wino": 3, "wino4": 4, "winoup": 5, "rowup": 6, "band": 7, "row": 8, "gemm_last": 9, "rowlast": 10}
FLAG_KEEP_INTERMEDIATES = 1
FLAG_NO_GRAPH = 2
FLAG_INSTANCE_NORM = 4
FLAG_WINO4 = 8
NORM_IDS = {"batch": 0, "instance": 1}


class NativeLibraryError(RuntimeError):
    pass


# ---- switches of tools, tests and A-B runs.  The LIBRARY never reads the process environment: the switches travel as the `tune` string of
# lspf2f_create_tuned ("key=value,..."; include/lspf2f.h lists the keys).  For shell-driven A-B scripts (tools/*.sh) the Python host maps
# LSP_HIP_<KEY>=<int> environment variables onto those keys HERE, once per Engine; an explicit `tune=` argument wins.
_PRESENCE_ENV = {   # legacy names whose mere presence is the switch
    "LSP_HIP_LASTCONV_DIRECT": ("lastconv_direct", 1), "LSP_HIP_LASTCONV_STRIP": ("lastconv", 1), "LSP_HIP_LASTCONV_ROWS": ("lastconv", 2),
    "LSP_HIP_LASTCONV_GENERIC": ("lastconv", 3), "LSP_HIP_LASTCONV_MFMA": ("lastconv", 4), "LSP_HIP_LASTCONV_VALU": ("lastconv", 5),
    "LSP_HIP_FIRSTCONV_DIRECT": ("firstconv", 1), "LSP_HIP_FIRSTCONV_REGSTAGE": ("firstconv", 2),
}
_RENAMED_ENV = {"LSP_HIP_XCD": "igemm_xcd", "LSP_HIP_FULLK_SPLIT_TILES": "fullk_split_tiles"}
# the keys lspf2f_create_tuned knows (include/lspf2f.h); an LSP_HIP_* variable that maps to none of them is a switch of the Python host
# (LSP_HIP_CAND_CACHE, networks.py) or a typo: it is not handed to the library (a typo gets a warning)
TUNE_KEYS = frozenset((
    "graph", "wino", "wino4", "wino_pre", "wino_ureg", "wino_prio", "in_wino_stats", "wino_xcd", "wino_il", "wino_rot", "winoup", "winoup_nb", "winoup_target", "igemm_xcd",
    "bandconv", "bandconv_min_blocks", "bandconv_min_frames", "patch16", "patchup16", "patch16_deep", "patch16_min_blocks", "tail_prefetch", "tail_prefetch_at", "tail_prefetch_wgs", "tail_prefetch_mb", "rowup", "rowlast", "rowlast_fused", "rowconv", "fullk_split", "fullk_split_tiles", "fullk_s2",
    "all_forms", "blob_pad_kb", "fused_splitk", "fused_splitk16", "fullk16", "fullk16_min_frames", "out_wt", "prefetch", "smallm_dma", "smallm_kb", "in_small_regs", "in_smallm_fused", "in_small_max_hw", "lastconv_direct", "lastconv", "firstconv"))
_HOST_ENV = frozenset(("LSP_HIP_CAND_CACHE", "LSP_HIP_TUNE"))


def tune_string(tune=None) -> bytes:
    """`tune` (None, a dict or a "k=v,k=v" string) merged over the LSP_HIP_* environment variables, as the library wants it."""
    kv = {}
    for name, val in os.environ.items():
        if not name.startswith("LSP_HIP_"):
            continue
        if name in _PRESENCE_ENV:
            k, v = _PRESENCE_ENV[name]
            kv.setdefault(k, v)
            continue
        if name in _HOST_ENV:
            continue
        key = _RENAMED_ENV.get(name, name[len("LSP_HIP_"):].lower())
        if key not in TUNE_KEYS:
            import warnings
            warnings.warn("%s is not a switch of liblspf2f (known keys: include/lspf2f.h): ignored" % name)
            continue
        try:
            kv[key] = int(val)
        except ValueError:
            raise ValueError("%s=%r: the switches of liblspf2f take integers" % (name, val))
    if "LSP_HIP_TUNE" in os.environ:
        tune_env = os.environ["LSP_HIP_TUNE"]
        for tok in filter(None, (t.strip() for t in tune_env.split(","))):
            k, v = tok.split("=")
            kv[k.strip()] = int(v)
    if isinstance(tune, str):
        for tok in filter(None, (t.strip() for t in tune.split(","))):
            k, v = tok.split("=")
            kv[k.strip()] = int(v)
    elif tune:
        kv.update({k: int(v) for k, v in tune.items()})
    return ",".join("%s=%d" % (k, v) for k, v in sorted(kv.items())).encode()


class Lspf2fError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("lspf2f error %d (%s): %s" % (code, ERR_NAMES.get(code, "?"), msg))
        self.code = code


class Config(Structure):
    _fields_ = [("abi_version", c_int32), ("variant", c_int32), ("input_nc", c_int32),
                ("feat_nc", c_int32), ("output_nc", c_int32), ("ngf", c_int32),
                ("num_downs", c_int32), ("height", c_int32), ("width", c_int32),
                ("max_batch", c_int32), ("dtype", c_int32), ("flags", c_uint32)]


class LayerInfo(Structure):
    _fields_ = [("name", c_char_p), ("kernel", c_char_p),
                ("cin", c_int32), ("cout", c_int32), ("h_in", c_int32), ("h_out", c_int32),
                ("stride", c_int32), ("upsample", c_int32), ("concat", c_int32),
                ("residual", c_int32), ("relu", c_int32), ("tanh_out", c_int32),
                ("tile_m", c_int32), ("tile_n", c_int32), ("split_k", c_int32), ("k_group", c_int32),
                ("flops_per_frame", c_int64), ("exec_flops_per_frame", c_int64), ("act_bytes_per_frame", c_int64),
                ("weight_bytes", c_int64), ("w_offset", c_int64), ("scale_offset", c_int64),
                ("shift_offset", c_int64), ("out_offset", c_int64)]


# every symbol include/lspf2f.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "lspf2f_create": (c_int, [POINTER(Config), POINTER(c_void_p)]),
    "lspf2f_create_tuned": (c_int, [POINTER(Config), c_char_p, POINTER(c_void_p)]),
    "lspf2f_destroy": (c_int, [c_void_p]),
    "lspf2f_last_error": (c_char_p, []),
    "lspf2f_abi_version": (c_int, []),
    "lspf2f_num_tensors": (c_int, [c_void_p]),
    "lspf2f_tensor_info": (c_int, [c_void_p, c_int, POINTER(c_char_p), POINTER(c_int64 * 4), POINTER(c_int)]),
    "lspf2f_set_tensor": (c_int, [c_void_p, c_char_p, c_void_p, c_size_t]),
    "lspf2f_packed_bytes": (c_size_t, [c_void_p]),
    "lspf2f_pack_weights": (c_int, [c_void_p, c_void_p, c_size_t]),
    "lspf2f_bind_weights": (c_int, [c_void_p, c_void_p, c_size_t]),
    "lspf2f_workspace_bytes": (c_size_t, [c_void_p, c_int]),
    "lspf2f_bind_workspace": (c_int, [c_void_p, c_void_p, c_size_t]),
    "lspf2f_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "lspf2f_set_candidates": (c_int, [c_void_p, c_void_p, c_void_p]),
    "lspf2f_forward_ex": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "lspf2f_num_layers": (c_int, [c_void_p]),
    "lspf2f_plan_batch": (c_int, [c_void_p, c_int]),
    "lspf2f_layer_info_get": (c_int, [c_void_p, c_int, POINTER(LayerInfo)]),
    "lspf2f_forward_timed": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p,
                                     POINTER(c_float)]),
    "lspf2f_subset_timed": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, POINTER(c_int), c_int,
                                    POINTER(c_float), POINTER(c_int)]),
    "lspf2f_conv3x3_scratch_bytes": (c_size_t, [c_int] * 13),
    "lspf2f_conv3x3": (c_int, [c_void_p] * 7 + [c_int] * 14 + [c_void_p, c_size_t, c_void_p]),
    "lspf2f_wino_chain_scratch_bytes": (c_size_t, [c_int] * 5),
    "lspf2f_wino_chain": (c_int, [c_int] + [POINTER(c_void_p)] * 6 + [POINTER(c_int)] + [c_int] * 5 + [c_void_p, c_size_t, c_void_p]),
    "lspf2f_unet_prepare": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int, c_void_p, c_void_p]),
    "lspf2f_pixel_shuffle": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "lspf2f_clock_probe": (c_int, [c_void_p, c_uint32, c_void_p]),
    "lspf2f_memcpy": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "lspf2f_layer_form_offset": (c_int64, [c_void_p, c_int, c_int]),
    "lspf2f_debug_poison": (c_int, [c_void_p, ctypes.c_ubyte, c_void_p, POINTER(c_uint32)]),
}



class A2HConfig(Structure):
    """lspa2h_config (include/lspa2h.h)"""
    _fields_ = [(n, c_int32) for n in ("abi_version", "residual_layers", "residual_blocks", "residual_channels",
                                       "dilation_channels", "skip_channels", "kernel_size", "input_channels",
                                       "cond_channels", "hidden_size", "ncenter", "ndim", "loss",
                                       "max_audio_frames")] + [("flags", c_uint32)]


A2H_ABI_VERSION = 1
A2H_LOSS_IDS = {"GMM": 0, "L2": 1}
A2H_FLAG_SINGLE_WORKGROUP = 1
A2H_FLAG_CONSECUTIVE_BLOCKS = 2
_GEN_ARGS = [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_float, c_int, c_void_p, c_int, c_void_p]
# every symbol include/lspa2h.h declares
A2H_SIGNATURES = {
    "lspa2h_create": (c_int, [POINTER(A2HConfig), POINTER(c_void_p)]),
    "lspa2h_destroy": (c_int, [c_void_p]),
    "lspa2h_last_error": (c_char_p, []),
    "lspa2h_abi_version": (c_int, []),
    "lspa2h_receptive_field": (c_int, [c_void_p]),
    "lspa2h_num_tensors": (c_int, [c_void_p]),
    "lspa2h_tensor_info": (c_int, [c_void_p, c_int, POINTER(c_char_p), POINTER(c_size_t)]),
    "lspa2h_set_tensor": (c_int, [c_void_p, c_char_p, c_void_p, c_size_t]),
    "lspa2h_packed_bytes": (c_size_t, [c_void_p]),
    "lspa2h_pack_weights": (c_int, [c_void_p, c_void_p, c_size_t]),
    "lspa2h_bind_weights": (c_int, [c_void_p, c_void_p, c_size_t]),
    "lspa2h_workspace_bytes": (c_size_t, [c_void_p]),
    "lspa2h_bind_workspace": (c_int, [c_void_p, c_void_p, c_size_t]),
    "lspa2h_generate": (c_int, _GEN_ARGS),
    "lspa2h_generate_timed": (c_int, _GEN_ARGS + [POINTER(c_float), POINTER(c_float)]),
    "lspa2h_debug_cond": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_int), POINTER(c_int)]),
    "lspa2h_status": (c_int, [c_void_p, c_void_p, POINTER(c_uint32)]),
    "lspa2h_sample_gmm": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p]),
}

# every symbol include/lsplle.h declares
LLE_SIGNATURES = {
    "lsplle_last_error": (c_char_p, []),
    "lsplle_knn_workspace_bytes": (c_size_t, [c_int, c_int]),
    "lsplle_knn": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "lsplle_solve": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                             c_float, c_void_p]),
}



class RNNConfig(Structure):
    """lsprnn_config (include/lsprnn.h)"""
    _fields_ = [(n, c_int32) for n in ("abi_version", "cell", "num_layers", "input_size", "hidden_size", "max_steps")] + [("flags", c_uint32)]


RNN_ABI_VERSION = 1
RNN_FLAG_PER_LAYER = 1
RNN_CELL_IDS = {"GRU": 0, "LSTM": 1}
# every symbol include/lsprnn.h declares
RNN_SIGNATURES = {
    "lsprnn_create": (c_int, [POINTER(RNNConfig), POINTER(c_void_p)]),
    "lsprnn_destroy": (c_int, [c_void_p]),
    "lsprnn_last_error": (c_char_p, []),
    "lsprnn_abi_version": (c_int, []),
    "lsprnn_num_tensors": (c_int, [c_void_p]),
    "lsprnn_tensor_info": (c_int, [c_void_p, c_int, POINTER(c_char_p), POINTER(c_size_t)]),
    "lsprnn_set_tensor": (c_int, [c_void_p, c_char_p, c_void_p, c_size_t]),
    "lsprnn_packed_bytes": (c_size_t, [c_void_p]),
    "lsprnn_pack_weights": (c_int, [c_void_p, c_void_p, c_size_t]),
    "lsprnn_bind_weights": (c_int, [c_void_p, c_void_p, c_size_t]),
    "lsprnn_workspace_bytes": (c_size_t, [c_void_p]),
    "lsprnn_bind_workspace": (c_int, [c_void_p, c_void_p, c_size_t]),
    "lsprnn_forward": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "lsprnn_status": (c_int, [c_void_p, c_void_p, POINTER(c_uint32)]),
    "lsprnn_linear": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
}

# every symbol include/lspraster.h declares
RASTER_SIGNATURES = {
    "lspraster_last_error": (c_char_p, []),
    "lspraster_edge_maps": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
}
RASTER_POINT_DTYPES = {"int32": 0, "float32": 1, "float64": 2}

# every symbol include/lspmel.h declares
MEL_SIGNATURES = {
    "lspmel_last_error": (c_char_p, []),
    "lspmel_num_windows": (c_int, [c_int64]),
    "lspmel_basis_floats": (c_size_t, []),
    "lspmel_make_basis": (c_int, [c_void_p, c_size_t]),
    "lspmel_workspace_bytes": (c_size_t, [c_int]),
    "lspmel_compute": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
}



class UnetConfig(Structure):
    """lspunet_config (include/lspunet.h)"""
    _fields_ = [(n, c_int32) for n in ("abi_version", "input_nc", "feat_nc", "output_nc", "ngf", "num_downs", "size", "max_batch", "dtype")] + [("flags", c_uint32)]


UNET_ABI_VERSION = 1
UNET_FLAG_NO_GRAPH = 1
_UNET_FWD = [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p]
# every symbol include/lspunet.h declares
UNET_SIGNATURES = {
    "lspunet_create": (c_int, [POINTER(UnetConfig), c_char_p, POINTER(c_void_p)]),
    "lspunet_destroy": (c_int, [c_void_p]),
    "lspunet_last_error": (c_char_p, []),
    "lspunet_abi_version": (c_int, []),
    "lspunet_num_tensors": (c_int, [c_void_p]),
    "lspunet_tensor_info": (c_int, [c_void_p, c_int, POINTER(c_char_p), POINTER(c_int64 * 4), POINTER(c_int)]),
    "lspunet_set_tensor": (c_int, [c_void_p, c_char_p, c_void_p, c_size_t]),
    "lspunet_packed_bytes": (c_size_t, [c_void_p]),
    "lspunet_pack_weights": (c_int, [c_void_p, c_void_p, c_size_t]),
    "lspunet_bind_weights": (c_int, [c_void_p, c_void_p, c_size_t]),
    "lspunet_workspace_bytes": (c_size_t, [c_void_p, c_int]),
    "lspunet_bind_workspace": (c_int, [c_void_p, c_void_p, c_size_t]),
    "lspunet_forward": (c_int, _UNET_FWD),
    "lspunet_forward_timed": (c_int, _UNET_FWD + [POINTER(c_float)]),
    "lspunet_num_launches": (c_int, [c_void_p, c_int]),
    "lspunet_launch_info": (c_int, [c_void_p, c_int, c_int, POINTER(c_char_p), POINTER(c_char_p), POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
}

_lib = None


def load() -> ctypes.CDLL:
    """Load liblspf2f.so and bind every declared symbol; raises NativeLibraryError loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C livespeechportraits_amd/csrc`). There is no CPU/PyTorch fallback."
            % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise NativeLibraryError("failed to load %s: %s" % (LIB_PATH, e)) from e
    for name, (res, args) in list(SIGNATURES.items()) + list(A2H_SIGNATURES.items()) + list(LLE_SIGNATURES.items()) + list(RNN_SIGNATURES.items()) + list(RASTER_SIGNATURES.items()) + list(MEL_SIGNATURES.items()) + list(UNET_SIGNATURES.items()):
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise NativeLibraryError("%s does not export %s" % (LIB_PATH, name)) from e
        fn.restype = res
        fn.argtypes = args
    if lib.lspf2f_abi_version() != ABI_VERSION:
        raise NativeLibraryError("ABI version mismatch: library %d, binding %d"
                                 % (lib.lspf2f_abi_version(), ABI_VERSION))
    if lib.lspa2h_abi_version() != A2H_ABI_VERSION:
        raise NativeLibraryError("lspa2h ABI version mismatch: library %d, binding %d"
                                 % (lib.lspa2h_abi_version(), A2H_ABI_VERSION))
    if lib.lspunet_abi_version() != UNET_ABI_VERSION:
        raise NativeLibraryError("lspunet ABI version mismatch: library %d, binding %d" % (lib.lspunet_abi_version(), UNET_ABI_VERSION))
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != OK:
        msg = load().lspf2f_last_error()
        raise Lspf2fError(rc, msg.decode() if msg else "")


class Lspa2hError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("lspa2h error %d: %s" % (code, msg))
        self.code = code


def check_a2h(rc: int) -> None:
    if rc != OK:
        msg = load().lspa2h_last_error()
        raise Lspa2hError(rc, msg.decode() if msg else "")


class LsplleError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("lsplle error %d: %s" % (code, msg))
        self.code = code


def check_lle(rc: int) -> None:
    if rc != OK:
        msg = load().lsplle_last_error()
        raise LsplleError(rc, msg.decode() if msg else "")


class LsprnnError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("lsprnn error %d: %s" % (code, msg))
        self.code = code


def check_rnn(rc: int) -> None:
    if rc != OK:
        msg = load().lsprnn_last_error()
        raise LsprnnError(rc, msg.decode() if msg else "")


class LsprasterError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("lspraster error %d: %s" % (code, msg))
        self.code = code


def check_raster(rc: int) -> None:
    if rc != OK:
        msg = load().lspraster_last_error()
        raise LsprasterError(rc, msg.decode() if msg else "")


class LspmelError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("lspmel error %d: %s" % (code, msg))
        self.code = code


def check_mel(rc: int) -> None:
    if rc != OK:
        msg = load().lspmel_last_error()
        raise LspmelError(rc, msg.decode() if msg else "")


class LspunetError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("lspunet error %d (%s): %s" % (code, ERR_NAMES.get(code, "?"), msg))
        self.code = code


def check_unet(rc: int) -> None:
    if rc != OK:
        msg = load().lspunet_last_error()
        raise LspunetError(rc, msg.decode() if msg else "")
