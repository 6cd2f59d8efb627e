"""Builds 2dimageto3dmodel_amd/lib/libm355.so (gfx950 HIP kernels + C-ABI) in-tree with hipcc.

    python 2dimageto3dmodel_amd/build.py [--force]

Cross-compiles without a GPU.  The .so is git-ignored but travels with the tree to the GPU box.
Per-file flags: the projection files are compiled with -ffp-contract=off because the bin index of a
point must be bit-exact with the torch-CPU reference (see csrc/proj_transform.hip).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build" + ("_" + os.environ["M355_BUILD_LIB"] if os.environ.get("M355_BUILD_LIB") else ""))
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, os.environ.get("M355_BUILD_LIB", "libm355.so"))  # A/B builds: M355_BUILD_LIB + M355_BUILD_DEFS
ARCH = "gfx950"

COMMON = ["-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-Wall", "-Wno-unused-function"] + os.environ.get("M355_BUILD_DEFS", "").split()
STRICT = ["-ffp-contract=off"]  # bit-exactness with torch-CPU elementwise arithmetic

SOURCES = [
    ("error.cpp", []),
    ("proj_transform.hip", STRICT),
    ("proj_render.hip", STRICT),
    ("proj_render21.hip", STRICT),
    ("sil_loss.hip", STRICT),
    ("proj_dense.hip", STRICT),
    ("chamfer.hip", STRICT),
    ("conv_mfma.hip", []),
    ("conv_small.hip", []),
    ("conv_halo.hip", []),
    ("conv_exact.hip", []),
    ("gan_elem.hip", []),
    ("gan_glue.hip", []),
    ("gan_io.hip", []),
    ("ipc_exchange.hip", []),
    ("mesh_deform.hip", STRICT),
    ("dibr_raster.hip", STRICT),
]


def source_hash():
    """sha256 over the kernel sources (csrc/*, include/m355.h), by sorted file name: ties a measured artefact (profiles/pmc_traffic.json:
    HBM bytes per launch from rocprofv3 PMC passes) to the code it was measured on -- bench.py nulls `roofline.traffic` when they differ"""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)) + [os.path.join(os.path.dirname(HERE), "include", "m355.h")]
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _newer(src, dst, extra=()):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(p) > t for p in (src,) + tuple(extra))


def build(force=False, verbose=True, exact=False):
    """exact=True: lib/libm355_exact.so -- the same sources with -DM355_EXACT (fp32 activations, fp32 convs with fp64
    accumulation, csrc/conv_exact.hip; include/m355.h m355_act_bytes): the library behind M355_EXACT=1 / _lib.set_exact(True)"""
    global OBJ, LIB, COMMON
    if exact:
        saved = (OBJ, LIB, COMMON)
        OBJ, LIB, COMMON = os.path.join(HERE, "build_exact"), os.path.join(LIBDIR, "libm355_exact.so"), COMMON + ["-DM355_EXACT"]
        try:
            return build(force, verbose)
        finally:
            OBJ, LIB, COMMON = saved
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    headers = tuple(os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")) + (
        os.path.join(os.path.dirname(HERE), "include", "m355.h"), os.path.abspath(__file__))
    cc = hipcc()
    objs = []
    procs = []
    for name, extra in SOURCES:
        src = os.path.join(CSRC, name)
        obj = os.path.join(OBJ, name.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        if force or _newer(src, obj, headers):
            cmd = [cc, f"--offload-arch={ARCH}", "-c", src, "-o", obj] + COMMON + extra
            if name.endswith(".cpp"):
                cmd = [cc, "-c", src, "-o", obj] + COMMON + extra
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((name, subprocess.Popen(cmd)))
    for name, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {name}")
    if force or procs or not os.path.exists(LIB):
        cmd = [cc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, exact="--exact" in sys.argv))
