"""ctypes binding of libta355.so (the C-ABI boundary, include/ta355.h).

* ``build()`` compiles every HIP source for gfx950 with hipcc into
  ``tiny_audio_amd/libta355.so`` (in-tree, so it travels with the repo snapshot).
* ``lib()`` loads it and binds EVERY prototype found in the header (argument and
  return types are parsed from the header text, so header and binding cannot drift).
* There is no fallback: if the shared object is absent ``lib()`` raises, and every
  op raises ``Ta355Error`` on a non-zero status.
"""
from __future__ import annotations

import ctypes as C
import glob
import os
import re
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HEADER = os.path.join(ROOT, "include", "ta355.h")
CSRC = os.path.join(HERE, "csrc")
SO_PATH = os.path.join(HERE, "libta355.so")
SOURCES = ["gemm.hip", "gemm_tn.hip", "norm.hip", "attention.hip", "attention_enc.hip", "qkv_post.hip", "elementwise.hip", "loss.hip",
           "logmel.hip", "optim.hip", "moe.hip", "lora.hip", "nn_prims.hip", "generate.hip", "decode_fused.hip", "api.hip"]


class Ta355Error(RuntimeError):
    pass


# ----------------------------------------------------------------------------- structs (must mirror ta355.h)
class EncLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ln1_w", "ln1_b", "ln2_w", "ln2_b", "wqkv", "bqkv", "wo", "bo",
                                          "w1", "b1", "w2", "b2", "wqkv_fa", "bqkv_fa")]


class EncoderWeights(C.Structure):
    _fields_ = [("hidden", C.c_int), ("ffn", C.c_int), ("n_layers", C.c_int), ("heads", C.c_int),
                ("n_mels", C.c_int), ("max_pos", C.c_int), ("ln_eps", C.c_float),
                ("conv1_w", C.c_void_p), ("conv1_b", C.c_void_p), ("conv2_w", C.c_void_p), ("conv2_b", C.c_void_p),
                ("norm_w", C.c_void_p), ("norm_b", C.c_void_p), ("rope_cos", C.c_void_p), ("rope_sin", C.c_void_p),
                ("layers", C.POINTER(EncLayer)), ("rope_il", C.c_void_p), ("res_f32", C.c_int)]


class MlpWeights(C.Structure):
    _fields_ = [("enc_dim", C.c_int), ("k", C.c_int), ("hidden", C.c_int), ("llm_dim", C.c_int), ("eps", C.c_float),
                ("w1", C.c_void_p), ("w2", C.c_void_p), ("w2_t", C.c_void_p), ("g1", C.c_void_p), ("g2", C.c_void_p)]


class MoeWeights(C.Structure):
    _fields_ = [("enc_dim", C.c_int), ("k", C.c_int), ("hidden", C.c_int), ("llm_dim", C.c_int), ("num_experts", C.c_int),
                ("eps", C.c_float), ("aux_coef", C.c_float), ("z_coef", C.c_float),
                ("norm_w", C.c_void_p), ("router_w", C.c_void_p),
                ("w1", C.POINTER(C.c_void_p)), ("w1_t", C.POINTER(C.c_void_p)), ("b1", C.POINTER(C.c_void_p)),
                ("w2", C.POINTER(C.c_void_p)), ("w2_t", C.POINTER(C.c_void_p)), ("b2", C.POINTER(C.c_void_p))]


class LmLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ln_in_w", "wqkv", "wqkv_t", "qn_w", "kn_w", "wo", "wo_t", "ln_post_w",
                                          "wgu", "wgu_t", "wd", "wd_t",
                                          "la_qkv", "lb_qkv", "la_o", "lb_o", "la_gu", "lb_gu", "la_d", "lb_d")]


class GemmOpts(C.Structure):
    _fields_ = [("a2", C.c_void_p), ("w2", C.c_void_p), ("k2", C.c_int), ("lda2", C.c_long), ("residual_bf16", C.c_void_p),
                ("rope_tab", C.c_void_p), ("rope_rows", C.c_int), ("w_blocked", C.c_int), ("rope_cols", C.c_int)]


class AttnLayout(C.Structure):
    _fields_ = [(n, C.c_long) for n in ("q_bs", "q_hs", "q_rs", "k_bs", "k_hs", "k_rs", "v_bs", "v_hs", "v_rs")]


class LmLoraGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("dla_qkv", "dlb_qkv", "dla_o", "dlb_o", "dla_gu", "dlb_gu", "dla_d", "dlb_d")]


class LmWeights(C.Structure):
    _fields_ = [("vocab", C.c_int), ("vocab_pad", C.c_int), ("hidden", C.c_int), ("ffn", C.c_int),
                ("n_layers", C.c_int), ("heads", C.c_int), ("kv_heads", C.c_int), ("head_dim", C.c_int),
                ("max_pos", C.c_int), ("eps", C.c_float),
                ("embed_f32", C.c_void_p), ("embed_bf16", C.c_void_p), ("embed_t_bf16", C.c_void_p),
                ("norm_w", C.c_void_p), ("rope_cos", C.c_void_p), ("rope_sin", C.c_void_p),
                ("layers", C.POINTER(LmLayer)), ("lora_rank", C.c_int), ("lora_scale", C.c_float), ("train_base", C.c_int),
                ("lora_groups", C.c_int), ("res_f32", C.c_int), ("dx_f32", C.c_int)]


class LmLayerWgrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("dwqkv", "dwo", "dwgu", "dwd", "dln_in", "dln_post", "dqn", "dkn")]


class LmWgrads(C.Structure):
    _fields_ = [("layers", C.POINTER(LmLayerWgrads)), ("dnorm", C.c_void_p), ("dembed", C.c_void_p)]


# ----------------------------------------------------------------------------- header parsing
_SCALARS = {"int": C.c_int, "long": C.c_long, "float": C.c_float, "double": C.c_double, "unsigned long long": C.c_ulonglong,
            "hipStream_t": C.c_void_p}


def _ctype(decl: str):
    decl = decl.strip()
    if "*" in decl:
        return C.c_void_p
    ty = re.sub(r"\b\w+$", "", decl).strip() if decl not in _SCALARS else decl   # drop the parameter name
    ty = ty.replace("const ", "").strip()
    if ty not in _SCALARS:
        raise ValueError(f"unmapped C type in ta355.h: {decl!r}")
    return _SCALARS[ty]


def parse_header(path: str = HEADER):
    """-> {name: (restype, [argtypes])} for every ``int|long ta_*(...)`` prototype."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|long)\s+(ta_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        argtypes = [] if args in ("", "void") else [_ctype(a) for a in args.split(",")]
        protos[name] = (_SCALARS[ret], argtypes)
    return protos


# ----------------------------------------------------------------------------- build / load
def hipcc_path():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 every kernel source into one shared object (cross-compiles without a GPU).
    Sources are compiled to objects in parallel (csrc/build/, only those older than their inputs) and then linked."""
    from concurrent.futures import ThreadPoolExecutor
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [HEADER]      # every header (gelu_lut.h / logmel_twiddles.h were missing)
    deps = srcs + hdrs
    if not force and os.path.exists(SO_PATH) and all(os.path.getmtime(SO_PATH) >= os.path.getmtime(d) for d in deps):
        return SO_PATH
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    # -amdgpu-mfma-vgpr-form: keep MFMA accumulators in VGPRs (gfx950's file is unified).  The default AGPR form made
    # the attention kernels shuttle every score / output fragment through v_accvgpr_read/write around the softmax.
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form", "-fPIC"]
    # attention_enc.hip: no NaN can reach its row maxima (scores are finite MFMA sums; masked keys are -1e30, not -inf), and
    # without this every fmaxf on an MFMA result is preceded by a canonicalising v_max_f32 x, x (12 extra VALU per key tile)
    extra = {"attention_enc.hip": ["-fno-honor-nans"]}
    no_vgpr_form = set()                # (sources that keep their accumulators in AGPRs on purpose: none in the product library)
    hdr_time = max(os.path.getmtime(h) for h in hdrs)

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), hdr_time):
            return obj, None
        fl = [f for f in flags if os.path.basename(src) not in no_vgpr_form or f not in ("-mllvm", "-amdgpu-mfma-vgpr-form")]
        cmd = [hipcc_path(), *fl, *extra.get(os.path.basename(src), []), "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        return obj, (None if r.returncode == 0 else r.stdout + r.stderr)

    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 4)) as pool:
        results = list(pool.map(compile_one, srcs))
    errs = [e for _, e in results if e]
    if errs:
        raise Ta355Error("hipcc failed:\n" + "\n".join(errs))
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", *[o for o, _ in results], "-o", SO_PATH]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise Ta355Error("hipcc link failed:\n" + r.stdout + r.stderr)
    return SO_PATH


_LIB = None

# Test instrumentation ONLY (tests/test_dryrun_plumbing.py): with DRY_RUN set, kernel-launching entry points are
# replaced by stubs that marshal their arguments through the real ctypes prototypes (catching arity / type /
# None-pointer mistakes in the Python plumbing on a GPU-less box) and then do NOTHING.  It computes no results and
# is never a fallback: nothing in the package sets it.
DRY_RUN = False


class _DryLib:
    def __init__(self, handle):
        self._h = handle
        self.calls = []

    def __getattr__(self, name):
        fn = getattr(self._h, name)
        if name.endswith(("_bytes", "_floats")) or name == "ta_version":      # host-only calls (sizes, the ABI version)
            return fn

        def stub(*args):
            if len(args) != len(fn.argtypes):
                raise TypeError(f"{name}: expected {len(fn.argtypes)} arguments, got {len(args)}")
            for i, (a, t) in enumerate(zip(args, fn.argtypes)):
                try:
                    t.from_param(a)
                except Exception as e:  # noqa: BLE001
                    raise TypeError(f"{name}: argument {i} ({a!r}) does not convert to {t.__name__}") from e
            self.calls.append(name)
            return 0
        return stub


def lib():
    """The loaded library with every header prototype bound.  Raises if libta355.so is missing."""
    global _LIB
    if DRY_RUN:
        if not isinstance(_LIB, _DryLib):
            _LIB = None
            real = _load()
            _LIB = _DryLib(real)
        return _LIB
    if isinstance(_LIB, _DryLib):
        _LIB = None
    return _load()


def _load():
    global _LIB
    if _LIB is None:
        if not os.path.exists(SO_PATH):
            raise Ta355Error(f"{SO_PATH} not found: run `python __graft_entry__.py` (build()) first. "
                             "There is no CPU fallback for the ta355 hot path.")
        # Load torch FIRST: its wheel bundles its own libamdhip64.  If libta355.so were loaded before it, the
        # process would end up with two HIP runtimes (ours from /opt/rocm, torch's from the wheel) and device
        # pointers / streams allocated by one would be invalid in the other (launch failures).
        import torch  # noqa: F401
        # TA355_LIB=<path>: load another build of the same ABI (A/B of two kernel versions inside one gpurun visit)
        handle = C.CDLL(os.environ.get("TA355_LIB") or SO_PATH)
        for name, (ret, argtypes) in parse_header().items():
            try:
                fn = getattr(handle, name)    # AttributeError if the header declares something not exported
            except AttributeError:
                if os.environ.get("TA355_LIB"):   # an older build under A/B may lack the newest entry points
                    continue
                raise
            fn.restype = ret
            fn.argtypes = argtypes
        _LIB = handle
    return _LIB


def check(status: int, what: str = "ta355 call"):
    if status != 0:
        raise Ta355Error(f"{what} failed with status {status} "
                         f"({ {1: 'bad argument', 2: 'kernel launch failure'}.get(status, 'unknown')})")
