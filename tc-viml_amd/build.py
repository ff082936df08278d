"""Builds the in-tree HIP library `tc-viml_amd/libtcv_hip.so` for gfx950 (MI355X).

    python tc-viml_amd/build.py [--force]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libtcv_hip.so")
SOURCES = ["tcv_capi.hip", "tcv_solve.hip", "tcv_marg.hip", "tcv_preint.hip", "tcv_gauge.hip", "tcv_lines.hip", "tcv_microbench.hip", "tcv_pack.cpp", "tcv_estimator.cpp"]
HEADERS = ["tcv_math.h", "tcv_gauge.h", "tcv_factors.h", "tcv_packed.h", "tcv_host.h", "tcv_dev.h", os.path.join("..", "..", "include", "tcv.h"), os.path.join("..", "..", "include", "tcv_estimator.h")]


def _stale() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    for f in SOURCES + HEADERS:
        p = os.path.join(CSRC, f)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


SANITIZE_FLAGS = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-gline-tables-only", "-shared-libsan"]


TSAN_FLAGS = ["-fsanitize=thread", "-fno-omit-frame-pointer", "-gline-tables-only", "-shared-libsan"]


def tsan_runtime() -> str:
    """the shared ThreadSanitizer runtime of the ROCm clang (preloaded by tests/test_sanitize_cpu.py, like the ASan one)"""
    import glob
    hits = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.tsan-x86_64.so"))
    return hits[-1] if hits else ""


def asan_runtime() -> str:
    """the shared AddressSanitizer runtime of the ROCm clang: a process that dlopens the sanitized library must preload it"""
    import glob
    hits = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    return hits[-1] if hits else ""


def build(force: bool = False, verbose: bool = False, profile: bool = False, ablate: bool = False, sanitize: bool = False) -> str:
    """profile=True builds libtcv_hip_prof.so with per-phase cycle accounting, ablate=True libtcv_hip_abl.so whose solve kernel can
    skip phases (TCV_ABLATE_SKIP bit mask): developer tools.  sanitize=True builds libtcv_hip_san.so: the HOST side (packer, C-ABI,
    native estimator, host halves of the .hip files) under AddressSanitizer + UndefinedBehaviorSanitizer -- the device pass ignores
    the flags (GPU sanitizers are not available on this pool); tests/test_sanitize_cpu.py drives it without a device."""
    # developer builds with arbitrary switches: python tc-viml_amd/build.py --suffix=lin3 -DTCV_ABLATE_CONST=0x... -DTCV_CHAIN_OCC=3  ->  libtcv_hip_lin3.so
    sfx = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--suffix=")]
    if sfx:
        return _compile(os.path.join(HERE, "libtcv_hip_%s.so" % sfx[0]), verbose, [a for a in sys.argv if a.startswith("-D")])
    if "--occ1" in sys.argv:      # developer A/B build: chain kernel at one wavefront per SIMD (no register spills), see tcv_solve.hip
        return _compile(os.path.join(HERE, "libtcv_hip_occ1.so"), verbose, ["-DTCV_CHAIN_OCC1=1"])
    if "--margocc1" in sys.argv:  # developer A/B build: marginalisation kernel at one wavefront per SIMD (no register spills), tools/r05_marg_spill_ab.sh
        return _compile(os.path.join(HERE, "libtcv_hip_margocc1.so"), verbose, ["-DTCV_MARG_OCC1=1"])
    if "--norefine" in sys.argv:  # developer A/B build: the Cholesky's panel solve without its refinement step (what it costs: profiles/r05_panel_refine_ab.txt)
        return _compile(os.path.join(HERE, "libtcv_hip_norefine.so"), verbose, ["-DTCV_PANEL_NOREFINE=1"])
    if "--occ3" in sys.argv:      # developer A/B build: chain kernel at three wavefronts per SIMD (168 registers), tools/dev_occupancy3.py
        return _compile(os.path.join(HERE, "libtcv_hip_occ3.so"), verbose, ["-DTCV_CHAIN_OCC3=1"])
    if "--tsan" in sys.argv or os.environ.get("TCV_BUILD_TSAN") == "1":      # ThreadSanitizer over the host side: worker pool, plan caches, block pool
        return _compile(os.path.join(HERE, "libtcv_hip_tsan.so"), verbose, TSAN_FLAGS, opt="-O1", link_extra=["-fsanitize=thread", "-shared-libsan"])
    if sanitize:
        return _compile(os.path.join(HERE, "libtcv_hip_san.so"), verbose, SANITIZE_FLAGS, opt="-O1", link_extra=["-fsanitize=address,undefined", "-shared-libsan"])
    if profile:
        return _compile(os.path.join(HERE, "libtcv_hip_prof.so"), verbose, ["-DTCV_PROFILE=1"])
    if ablate:
        return _compile(os.path.join(HERE, "libtcv_hip_abl.so"), verbose, ["-DTCV_ABLATE=1"])
    if not force and not _stale():
        return OUT
    return _compile(OUT, verbose, [])


def _compile(out: str, verbose: bool, extra, opt: str = "-O3", link_extra=()) -> str:
    """One object per source (compiled in parallel), then one link.  tcv_solve.hip is compiled twice: the dense kernels, and
    (-DTCV_SOLVE_CHAIN_TU) the chain kernel alone in its own translation unit so that its 2-waves-per-SIMD register budget
    is not widened by the other instantiations' call trees."""
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    base = [hipcc, "--offload-arch=gfx950", opt, "-std=c++17", "-fPIC", "-ffp-contract=on",
            "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-value", "-Wno-unused-result", "-Wno-option-ignored", "-x", "hip"]
    if os.path.exists(os.path.join(CSRC, "tcv_marg.hip")):
        base.append("-DTCV_HAVE_MARG=1")
    if verbose:
        base.append("-Rpass-analysis=kernel-resource-usage")
    base += list(extra)
    objdir = os.path.join(HERE, "build", os.path.basename(out).replace(".so", ""))
    os.makedirs(objdir, exist_ok=True)
    jobs = [(f, os.path.join(objdir, f + ".o"), []) for f in SOURCES if os.path.exists(os.path.join(CSRC, f))]
    # (-DTCV_CAMW_CONST: the width of the camera-space vectors as a literal -- windows with a relocalisation pose, whose vectors are wider,
    # run on the ProjectionTdFactor instance below, tcv_capi.hip tcv_batch_solve; the cooperative mode does not take them)
    jobs.append(("tcv_solve.hip", os.path.join(objdir, "tcv_solve_chain.o"), ["-DTCV_SOLVE_CHAIN_TU=1", "-DTCV_CAMW_CONST=1"]))
    jobs.append(("tcv_solve.hip", os.path.join(objdir, "tcv_solve_coop.o"), ["-DTCV_SOLVE_COOP_TU=1", "-DTCV_CAMW_CONST=1"]))      # the cooperative small-batch kernel
    jobs.append(("tcv_solve.hip", os.path.join(objdir, "tcv_solve_chain_td.o"), ["-DTCV_SOLVE_CHAIN_TD_TU=1"]))      # the chain kernel with ProjectionTdFactor

    def one(job):
        src, obj, flags = job
        subprocess.check_call(base + flags + ["-c", os.path.join(CSRC, src), "-o", obj])
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
        objs = list(ex.map(one, jobs))
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + list(link_extra) + objs + ["-o", out])
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv, profile="--profile" in sys.argv, ablate="--ablate" in sys.argv, sanitize="--sanitize" in sys.argv))
