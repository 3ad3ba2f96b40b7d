"""ctypes loader of libhssfsst.so (C ABI: include/hssfsst.h).

The shared library is built in-tree by ``build()`` (hipcc --offload-arch=gfx950) and is the only
compute implementation of this package: if it is missing or fails to load, every entry point raises
-- there is no Python/CPU fallback.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import threading

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libhssfsst.so")
SRC = os.path.join(_PKG, "csrc", "hssfsst.hip")
HEADER = os.path.join(os.path.dirname(_PKG), "include", "hssfsst.h")

MODE_RAW, MODE_ABS, MODE_STACK, MODE_STACK_UNNORM = 0, 1, 2, 3
E_INVAL, E_NODEVICE, E_UNSUPPORTED, E_NOMEM, E_HIP = -1, -2, -3, -4, -5

_lock = threading.Lock()
_lib = None
_fork_warned = False
hip_owner_pid = None     # pid of the process in which this package first touched HIP (inherited by forked children)


class ForkedAfterGpuInitError(RuntimeError):
    """A DataLoader worker forked after its parent initialised the GPU tried to use the transform (a RuntimeError, as the
    reference's dataset expects, but also logged at ERROR and warned once: see ``guard_fork``)."""


def guard_fork() -> None:
    """HIP cannot be used in a child forked AFTER the parent initialised it (the reference forks DataLoader workers,
    /root/reference/main.py:202-218).  Creating the plan lazily in the worker is fine -- that is the supported pattern;
    a worker forked from a parent that already ran a transform gets a RuntimeError here (the exception the reference's
    dataset catches, hss/datasets/heart_sounds.py:183) instead of undefined behaviour inside the driver."""
    global hip_owner_pid
    pid = os.getpid()

    def refuse(msg: str):
        global _fork_warned
        # The reference's dataset catches RuntimeError, prints it and returns None (heart_sounds.py:183): with
        # in_memory=False and num_workers > 0 every sample of such a worker becomes None and the failure surfaces later, in
        # collate_fn.  Say it once, loudly, where it happens.
        import logging
        import warnings
        if not _fork_warned:                               # once per process: with in_memory=False every sample of the worker gets here
            _fork_warned = True
            logging.getLogger("heart_sounds_segmentation_amd").error(msg)
            warnings.warn(msg, RuntimeWarning, stacklevel=3)
        raise ForkedAfterGpuInitError(msg)

    if hip_owner_pid is not None and hip_owner_pid != pid:
        refuse("FSST: the HIP runtime was initialised in the parent process before this worker was forked; "
               "create / first use the transform inside the worker (lazy plan), or start workers with "
               "multiprocessing_context='spawn'")
    try:
        import torch
        bad = torch.cuda._is_in_bad_fork()
    except (ImportError, AttributeError):
        bad = False
    if bad:
        refuse("FSST: torch initialised the GPU in the parent process before this worker was forked; "
               "use multiprocessing_context='spawn' or touch the GPU only inside the workers")
    hip_owner_pid = pid


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/hssfsst.hip for gfx950 into libhssfsst.so (cross-compiles without a GPU)."""
    csrc = os.path.join(_PKG, "csrc")
    # every source under csrc/ (the MFMA core, the generic kernels, the host helpers) plus the C header
    deps = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".hpp", ".h"))) + [HEADER]
    if (not force and os.path.exists(LIB_PATH)
            and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps)):
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-o", LIB_PATH, SRC]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed ({res.returncode}): {' '.join(cmd)}\n{res.stdout}")
    return LIB_PATH


def check_code_object(path: str = None) -> dict:
    """The SHIPPED code object, looked at: fsst_team16_kernel keeps its held images in v104..v127 through inline assembly and compiles for 104
    allocatable registers (fsst_team16.hpp) -- a compiler that stopped honouring the limit, started using AGPRs or outlined a call would corrupt
    features silently.  Disassembles the gfx950 code object inside the library and asserts, for every team kernel: the only instructions
    that name a register >= v104 are the two accessors, all twelve pairs are used, no AGPR, no call, 128 registers reserved.  Returns counts."""
    import re
    import shutil
    import tempfile
    path = path or LIB_PATH
    llvm = "/opt/rocm/lib/llvm/bin"
    tools = {t: (shutil.which(t) or os.path.join(llvm, t)) for t in ("clang-offload-bundler", "llvm-objdump", "llvm-readelf")}
    objcopy = shutil.which("objcopy") or shutil.which("llvm-objcopy") or os.path.join(llvm, "llvm-objcopy")
    for t in list(tools.values()) + [objcopy]:
        if not os.path.exists(t):
            raise RuntimeError(f"check_code_object: {t} not found")
    with tempfile.TemporaryDirectory() as td:
        fat, obj = os.path.join(td, "fat.bin"), os.path.join(td, "code.o")
        subprocess.run([objcopy, "-O", "binary", "--only-section=.hip_fatbin", path, fat], check=True, capture_output=True)
        subprocess.run([tools["clang-offload-bundler"], "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--output={obj}"], check=True, capture_output=True)
        dis = subprocess.run([tools["llvm-objdump"], "-d", "--mcpu=gfx950", obj], check=True, capture_output=True, text=True).stdout.split("\n")
        notes = subprocess.run([tools["llvm-readelf"], "--notes", obj], check=True, capture_output=True, text=True).stdout
    starts = [i for i, ln in enumerate(dis) if re.match(r"^[0-9a-f]+ <_ZN7hssfsst18fsst_team16_kernel.*>:", ln)]
    if len(starts) < 2:
        raise RuntimeError("check_code_object: the team kernels are not in the code object")
    put = re.compile(r"^\s*v_pk_mul_f32 v\[(\d+):(\d+)\], v\[\d+:\d+\], v\[\d+:\d+\]\s*$")
    get = re.compile(r"^\s*v_pk_add_f32 v\[\d+:\d+\], v\[(\d+):(\d+)\], v\[\d+:\d+\] op_sel")
    out = {"kernels": 0, "accessor_instructions": 0}
    for st in starts:
        seen = set()
        for ln in dis[st + 1:]:
            if re.match(r"^[0-9a-f]+ <.*>:", ln):
                if "_ZN7hssfsst" in ln and "fsst_team16_kernel" not in ln:
                    break
                continue
            code = ln.split("//")[0]
            if re.search(r"\bs_swappc_b64\b|\bs_call_b64\b|\bs_setpc_b64\b", code):
                raise RuntimeError(f"check_code_object: a call inside the team kernel: {code.strip()}")
            if re.search(r"\ba\d+\b|\ba\[\d+:\d+\]|v_accvgpr", code):
                raise RuntimeError(f"check_code_object: an AGPR inside the team kernel: {code.strip()}")
            regs = [int(a) for a in re.findall(r"\bv(\d+)\b", code)] + [int(b) for _, b in re.findall(r"\bv\[(\d+):(\d+)\]", code)]
            if not regs or max(regs) < 104:
                continue
            m = put.match(code) or get.match(code)
            if not (m and int(m.group(1)) >= 104 and int(m.group(2)) == int(m.group(1)) + 1):
                raise RuntimeError(f"check_code_object: register >= v104 outside the accessors: {code.strip()}")
            if sum(1 for _, b in re.findall(r"\bv\[(\d+):(\d+)\]", code) if int(b) >= 104) != 1:
                raise RuntimeError(f"check_code_object: two held pairs in one instruction: {code.strip()}")
            seen.add(int(m.group(1)))
            out["accessor_instructions"] += 1
        if seen != set(range(104, 128, 2)):
            raise RuntimeError(f"check_code_object: held pairs used: {sorted(seen)}")
        out["kernels"] += 1
    # the kernels' metadata: 128 vector registers, no AGPR
    for blk in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
        if "fsst_team16_kernel" not in blk.split(".symbol:")[0] + blk:
            continue
        name = re.search(r"\.name:\s+(\S+)", blk)
        if not name or "fsst_team16_kernel" not in name.group(1):
            continue
        agpr = int(blk.strip().split()[0])
        vg = re.search(r"\.vgpr_count:\s+(\d+)", blk)
        if agpr != 0 or not vg or int(vg.group(1)) != 128:
            raise RuntimeError(f"check_code_object: {name.group(1)}: agpr_count {agpr}, vgpr_count {vg.group(1) if vg else None}")
        out["metadata_checked"] = out.get("metadata_checked", 0) + 1
    if out.get("metadata_checked", 0) < 2:
        raise RuntimeError("check_code_object: the team kernels' metadata was not found")
    return out


def lib():
    """The loaded C-ABI library; raises RuntimeError when it is absent (no fallback)."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension is the only implementation of the FSST "
                "path (no CPU fallback). Build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or heart_sounds_segmentation_amd._lib.build().")
        try:
            L = ctypes.CDLL(LIB_PATH)
        except OSError as e:  # e.g. libamdhip64 not found
            raise RuntimeError(f"cannot load {LIB_PATH}: {e}") from e
        c_int, c_dbl, c_i64 = ctypes.c_int, ctypes.c_double, ctypes.c_int64
        vp, ip = ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)
        dp = ctypes.POINTER(ctypes.c_double)
        L.hssfsst_plan_create.argtypes = [ctypes.POINTER(vp), c_int, c_int, dp, c_dbl, c_int, c_dbl, c_dbl, c_int]
        L.hssfsst_plan_create.restype = c_int
        L.hssfsst_plan_destroy.argtypes = [vp]
        L.hssfsst_plan_destroy.restype = c_int
        L.hssfsst_plan_info.argtypes = [vp, ip, ip, ip, ip, ip, ip, ip]
        L.hssfsst_plan_info.restype = c_int
        L.hssfsst_exec.argtypes = [vp, vp, c_i64, c_int, c_int, vp, c_int, vp]
        L.hssfsst_exec.restype = c_int
        L.hssfsst_exec_cols.argtypes = [vp, vp, c_i64, c_int, c_int, c_int, c_int, vp, c_int, vp]
        L.hssfsst_exec_cols.restype = c_int
        L.hssfsst_exec_frames.argtypes = [vp, vp, c_i64, c_int, c_i64, c_int, c_int, c_int, vp, c_int, vp]
        L.hssfsst_exec_frames.restype = c_int
        L.hssfsst_exec_pinned.argtypes = [vp, vp, c_int, ctypes.POINTER(vp)]
        L.hssfsst_exec_pinned.restype = c_int
        L.hssfsst_pinned_release.argtypes = [vp, vp]
        L.hssfsst_pinned_release.restype = c_int
        L.hssfsst_normalize_running.argtypes = [vp, vp, c_i64, c_int, vp, vp]
        L.hssfsst_normalize_running.restype = c_int
        L.hssfsst_exec_list.argtypes = [vp, vp, c_i64, vp, c_int, c_i64, c_int, c_int, vp, c_int, vp]
        L.hssfsst_exec_list.restype = c_int
        L.hssfsst_stream_step.argtypes = [vp, vp, c_i64, c_i64, vp, c_i64, c_int, c_int, c_int, vp, vp, vp, vp]
        L.hssfsst_stream_step.restype = c_int
        L.hssfsst_plan_last_exec_fused.argtypes = [vp]
        L.hssfsst_plan_set_zpath.argtypes = [vp, c_int]
        L.hssfsst_plan_fallbacks.argtypes = [vp]
        L.hssfsst_allgather.argtypes = [vp, vp, c_i64, vp, vp, c_int]
        L.hssfsst_allgather.restype = c_int
        L.hssfsst_plan_last_kernel.argtypes = [vp, ctypes.c_char_p, c_int]
        L.hssfsst_plan_last_kernel.restype = c_int
        L.hssfsst_plan_fallbacks.restype = c_int
        L.hssfsst_plan_set_zpath.restype = c_int
        L.hssfsst_plan_last_exec_fused.restype = c_int
        L.hssfsst_plan_check.argtypes = [vp]
        L.hssfsst_plan_check.restype = c_int
        L.hssfsst_plan_set_timing.argtypes = [vp, c_int]
        L.hssfsst_plan_set_timing.restype = c_int
        L.hssfsst_plan_timing.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ip]
        L.hssfsst_plan_timing.restype = c_int
        L.hssfsst_dtwin.argtypes = [dp, c_int, c_dbl, dp]
        L.hssfsst_dtwin.restype = c_int
        L.hssfsst_band.argtypes = [c_int, c_dbl, c_dbl, c_dbl, ip, ip]
        L.hssfsst_band.restype = c_int
        L.hssfsst_update_mean.argtypes = [c_dbl, c_dbl, c_i64]
        L.hssfsst_update_mean.restype = c_dbl
        L.hssfsst_update_variance.argtypes = [c_dbl, c_dbl, c_dbl, c_i64]
        L.hssfsst_update_variance.restype = c_dbl
        L.hssfsst_moments_merge.argtypes = [vp, vp, c_i64, c_int, vp, vp]
        L.hssfsst_moments_merge.restype = c_int
        L.hssfsst_parse_signal_csv.argtypes = [ctypes.c_char_p, c_i64, vp, vp, c_i64]
        L.hssfsst_parse_signal_csv.restype = c_i64
        L.hssfsst_pack_recordings.argtypes = [vp, vp, c_i64, c_int, c_int, vp, c_i64, vp, c_i64, c_int]
        L.hssfsst_pack_recordings.restype = c_i64
        L.hssfsst_resample.argtypes = [dp, c_i64, c_i64, dp]
        L.hssfsst_resample.restype = c_int
        L.hssfsst_device_count.restype = c_int
        L.hssfsst_version.restype = c_int
        L.hssfsst_last_error.restype = ctypes.c_char_p
        _lib = L
        return _lib


def check(rc: int, what: str) -> None:
    """Status -> exception: bad arguments raise ValueError, everything else -- including a
    configuration the kernels cannot run (E_UNSUPPORTED) -- RuntimeError, the one exception the
    reference's dataset catches around the transform (hss/datasets/heart_sounds.py:183)."""
    if rc == 0:
        return
    msg = lib().hssfsst_last_error().decode("utf-8", "replace")
    if rc == E_INVAL:
        raise ValueError(f"{what}: {msg} (status {rc})")
    raise RuntimeError(f"{what}: {msg} (status {rc})")
