"""Builds libpwclo_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python -m pwclonet_pylidarslam_amd.build [--force]

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting
``pwclonet_pylidarslam_amd/lib/libpwclo_hip.so`` travels to the GPU box with the tree (it is
git-ignored, not gpurun-ignored).  ``-ffp-contract=off``: the parity contract is source-order
IEEE fp32 for every distance / interpolation expression (SURVEY.md section 7 "Hard parts").
``-fno-slp-vectorize``: the SLP pass packs neighbouring fp32 adds into ``v_pk_add_f32``, which
blocks the DPP fusion of the K-neighbour reductions (2 ``v_mov_b32_dpp`` + 1 packed add instead
of 2 ``v_add_f32_dpp``) and is no faster than two scalar adds on gfx950.
"""
import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "lib", "obj")
LIB = os.path.join(HERE, "lib", "libpwclo_hip.so")
LIB_TRACE = os.path.join(HERE, "lib", "libpwclo_hip_trace.so")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libpwclo_hip.so")
    return exe


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _newest_header_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "pwclo_ops.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src, force, trace=False):
    obj = os.path.join(OBJ + ("_trace" if trace else ""), os.path.basename(src)[:-4] + ".o")
    stale = force or not os.path.exists(obj) or os.path.getmtime(obj) < max(
        os.path.getmtime(src), _newest_header_mtime())
    if stale:
        cmd = [hipcc()] + FLAGS + (["-DPWCLO_TRACE"] if trace else []) + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj, stale


def source_stamp():
    """sha256 (first 16 hex digits) over everything that decides what the FUSED FORWARD's kernels do and how much memory
    they touch -- the path the PMC artefacts describe: the stack / sampler / neighbour-search / warp sources of csrc/, their
    shared headers and the host-side packing / launch code (fused.py).  The training-only kernels
    (conv1x1, batchnorm) and the stand-alone ext ops are not part of it.  Profile artefacts that bench.py reports beside
    live timings (profiles/pmc_*.json) carry the stamp of the tree they were measured on; bench.py marks them stale when
    it differs from the running tree's."""
    import hashlib
    h = hashlib.sha256()
    names = ("common.hpp", "mlp_core.hpp", "fused_hoisted.hip", "fused_layers.hip", "fused_sa.hip", "knn.hip",
             "sampling.hip", "warp.hip", "state.hip")
    files = [os.path.join(CSRC, f) for f in names] + [os.path.join(HERE, "fused.py")]   # (the ABI header only declares)
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def build(force=False, jobs=None, trace=False):
    """trace=True builds the developer variant lib/libpwclo_hip_trace.so (-DPWCLO_TRACE: per-workgroup trace hooks,
    tools/wgtrace.py); the product library never carries them."""
    os.makedirs(OBJ + ("_trace" if trace else ""), exist_ok=True)
    lib = LIB_TRACE if trace else LIB
    srcs = sources()
    jobs = jobs or min(len(srcs), max(1, (os.cpu_count() or 2) - 1))
    with concurrent.futures.ThreadPoolExecutor(jobs) as ex:
        results = list(ex.map(lambda s: _compile(s, force, trace), srcs))
    objs = [o for o, _ in results]
    if force or any(st for _, st in results) or not os.path.exists(lib):
        cmd = [hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, trace="--trace" in sys.argv))
