"""Build libawr_hip.so with hipcc for gfx950 (in-tree, so the .so travels with the repo snapshot).

    python -m awr_amd.build            # or: __graft_entry__.build()

There is deliberately no fallback: if the library is missing and cannot be built, importing the
product path raises.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libawr_hip.so")
ARCH = "gfx950"

# per-source extra flags; the head/GT-map kernels must not contract a*b+c (see awr_head.hip)
SOURCES = {
    "awr_head.hip": ["-ffp-contract=off"],
    "awr_elem.hip": [],
    "awr_gemm_t22.hip": [],   # LDS-DMA implicit GEMM, one workgroup-tile shape per file (they dominate the build: compiled side by side)
    "awr_gemm_t21.hip": [],
    "awr_gemm_t12.hip": [],
    "awr_gemm_t11.hip": [],
    "awr_conv.hip": [],       # conv dispatch, register-staged GEMM, fused pairs
    "awr_wgrad.hip": [],      # weight gradients
    "awr_wino.hip": [],       # Winograd F(2x2, 3x3) forward of the stride-1 3x3 convolutions
    "awr_stem.hip": [],
    "awr_net.hip": [],        # host-only: network-level plan builder / runner
    "awr_dp.hip": [],         # host-only: RCCL communicators through dlopen (no link-time dependency)
    "awr_nyu.hip": ["-ffp-contract=off"],     # NYU data path: numpy's / OpenCV's arithmetic, no fused multiply-adds
}


# translation units of study builds only (AWR_BUILD_STUDY=1): measured, not adopted
STUDY_SOURCES = {}


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: cannot build libawr_hip.so")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=True, study=None):
    """study (default: $AWR_BUILD_STUDY == "1"): compile with -DAWR_STUDY -- the measured-and-rejected forms of earlier rounds (pre-cut
    split-operand GEMM, half-batch BatchNorm-backward wavefront) and their entry points.  The default library has none of them."""
    study = (os.environ.get("AWR_BUILD_STUDY") == "1") if study is None else bool(study)
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "awr_hip.h"))
    objs, jobs = [], []
    sources = dict(SOURCES)
    if study:
        sources.update(STUDY_SOURCES)
    for stale_obj in (os.path.join(LIBDIR, s.replace(".hip", ".o")) for s in STUDY_SOURCES):
        if not study and os.path.exists(stale_obj):
            os.remove(stale_obj)            # (an object of an earlier study build must not be linked into the default library)
    marker = os.path.join(LIBDIR, ".study")
    if os.path.exists(marker) != study:      # switching between the two kinds of build recompiles everything
        force = True
    for src, extra in sources.items():
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            continue
        obj = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        deps = [path] + headers + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".inc")]
        if force or _stale(obj, deps):
            # --offload-compress: the gfx950 code objects travel compressed inside the fat binary (the ~420 live instantiations of the LDS-DMA GEMM are
            # 10.5 MB of machine code uncompressed, 2 MB compressed; the HIP runtime inflates a translation unit's bundle when it loads it).
            # AWR_BUILD_NO_COMPRESS=1 builds without it.
            comp = [] if os.environ.get("AWR_BUILD_NO_COMPRESS") == "1" else ["--offload-compress"]
            jobs.append([hipcc, "-x", "hip", "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
                         "-munsafe-fp-atomics", "-c", path, "-o", obj] + comp + extra + (["-DAWR_STUDY"] if study else []))
        objs.append(obj)
    if jobs:            # translation units are independent: compile them side by side (the GEMM files dominate the wall time)
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print("[awr build]", " ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(run, jobs))
    if study:
        open(marker, "w").close()
    elif os.path.exists(marker):
        os.remove(marker)
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        if verbose:
            print("[awr build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv))
