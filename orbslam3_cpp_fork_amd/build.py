"""Build recipe of liborbx.so (HIP kernels + C ABI) for gfx950.

hipcc cross-compiles without a GPU.  The library is built in-tree (orbslam3_cpp_fork_amd/liborbx.so) so
that it travels with the repo snapshot to the GPU box.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "liborbx.so")
HIP_SOURCES = ["orbx_extractor.hip", "orbx_matcher.hip", "orbx_vocab.hip", "orbx_calib.hip", "orbx_comm.hip"]
HOST_SOURCES = ["orbx_hostmodel.cpp"]
HEADERS = ["orbx_kernels.h", "orbx_fast_tiles.h", "orbx_geom.h", "octree_core.h", "orb_sincosf_port.h", "orb_pattern.inc", "../../include/orbx.h"]
# -ffp-contract=off: float expressions must evaluate as the reference writes them (no FMA fusion).
# Division/sqrt stay correctly rounded (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt); no fast-math.
HIPFLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]
CXXFLAGS = ["-O2", "-ffp-contract=off", "-std=c++17", "-fPIC", "-Wall"]


STAMP = os.path.join(HERE, "_build", "liborbx.sha256")


def _tuning():
    """ORBX_BUILD_TUNING=1 compiles the tuning hooks in (-DORBX_TUNING: ORBX_DBG_SKIP_FAST phase ablation, ORBX_DBG_OCT stamps)."""
    return os.environ.get("ORBX_BUILD_TUNING", "") not in ("", "0")


def source_digest():
    """sha256 over every source, header, this recipe and the flags: what the in-tree liborbx.so must have been built from."""
    import hashlib

    h = hashlib.sha256()
    for f in sorted(HIP_SOURCES + HOST_SOURCES + HEADERS):
        path = os.path.join(CSRC, f)
        if os.path.exists(path):
            h.update(f.encode() + b"\0" + open(path, "rb").read())
    h.update(open(os.path.abspath(__file__), "rb").read())
    h.update(repr((HIPFLAGS, CXXFLAGS, _tuning(), os.environ.get("ORBX_TUNING_DEFINES", "") if _tuning() else "")).encode())
    return h.hexdigest()


def _stale():
    """By content, not by mtime: the .so is git-ignored but travels with the snapshot, and a checkout or copy can leave a source file older
    than a library that was built from something else."""
    if not os.path.exists(OUT) or not os.path.exists(STAMP):
        return True
    return open(STAMP).read().strip() != source_digest()


def build(force=False, verbose=False):
    if not force and not _stale():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = (["-DORBX_TUNING"] + os.environ.get("ORBX_TUNING_DEFINES", "").split()) if _tuning() else []  # e.g. ORBX_TUNING_DEFINES="-DORBX_MF_VARIANT=2"
    bdir = os.path.join(HERE, "_build")
    os.makedirs(bdir, exist_ok=True)
    objs = []
    for src in HOST_SOURCES:
        obj = os.path.join(bdir, os.path.splitext(src)[0] + ".o")
        cmd = ["g++"] + CXXFLAGS + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(obj)
    for src in HIP_SOURCES:
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            continue
        obj = os.path.join(bdir, os.path.splitext(src)[0] + ".o")
        cmd = [hipcc] + HIPFLAGS + extra + ["-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(obj)
    # Link with g++ against a generated stub of libamdhip64 (plain names, SONAME "libamdhip64.so") instead of letting
    # hipcc pin "libamdhip64.so.7" + symbol versions: PyTorch bundles its own HIP runtime, and two HIP runtimes in one
    # process cannot both open the GPU.  With the stub, the loader resolves "libamdhip64.so" through DT_RPATH (not RUNPATH: it must win over LD_LIBRARY_PATH, as torch's own $ORIGIN rpath does):
    # the PyTorch-bundled runtime when PyTorch is installed (shared with torch in-process, whatever the import order),
    # else /opt/rocm/lib.  ORBX_HIP_LIBDIR overrides the search order.
    syms = set()
    for obj in objs:
        out = subprocess.check_output(["nm", "--undefined-only", obj], text=True)
        for line in out.splitlines():
            name = line.split()[-1]
            if name.startswith("hip") or name.startswith("__hip"):
                syms.add(name.split("@")[0])
    stubdir = os.path.join(bdir, "hipstub")
    os.makedirs(stubdir, exist_ok=True)
    stub_c = os.path.join(stubdir, "stub.c")
    with open(stub_c, "w") as f:
        f.write("/* generated: link-time stand-in for the HIP runtime's exported names (never loaded at run time) */\n")
        for name in sorted(syms):
            f.write("void %s(void) {}\n" % name)
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-Wl,-soname,libamdhip64.so", "-o", os.path.join(stubdir, "libamdhip64.so"), stub_c])
    rpaths = []
    if os.environ.get("ORBX_HIP_LIBDIR"):
        rpaths.append(os.environ["ORBX_HIP_LIBDIR"])
    try:
        import importlib.util

        spec = importlib.util.find_spec("torch")
        if spec and spec.submodule_search_locations:
            tl = os.path.join(list(spec.submodule_search_locations)[0], "lib")
            if os.path.exists(os.path.join(tl, "libamdhip64.so")):
                rpaths.append(tl)
    except Exception:
        pass
    rpaths.append("/opt/rocm/lib")
    cmd = ["g++", "-shared", "-fPIC", "-o", OUT] + objs + ["-L" + stubdir, "-lamdhip64", "-lpthread", "-ldl"] + ["-Wl,--disable-new-dtags"] + ["-Wl,-rpath," + r for r in rpaths]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(source_digest() + "\n")
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
