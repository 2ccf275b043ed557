"""Build the HIP extension (libair_hip.so) in-tree for gfx950.

``python -m asvspoof2021_air_amd.build`` or ``__graft_entry__.build()``.
hipcc cross-compiles without a GPU; the .so stays in the tree (git-ignored) so
it travels to the GPU box with the snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(PKG, "_build")
LIBDIR = os.path.join(PKG, "_lib")
LIB = os.path.join(LIBDIR, "libair_hip.so")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CFLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
          "-I" + CSRC, "-Wall", "-Wno-unused-function", "-Wno-inline-asm", "-ffp-contract=off"]
# -ffp-contract=off: the oracle is plain fp32; kernels that want FMAs say fmaf().


# per-file extra flags (A/B knobs for single kernels)
# conv_wino4.hip: the SLP vectoriser packs the Winograd transforms into v_pk_* plus register shuffles,
# which is slower beside the f32 MFMAs than the scalar form (MI355X_MICROARCH.md, packed f32 VALU)
EXTRA = {"conv_wino4.hip": ["-fno-slp-vectorize"]}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(verbose=True, force=False):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(ROOT, "include", "air_hip.h"))
    jobs = []
    objs = []
    for src in _sources():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src[:-4] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([HIPCC] + CFLAGS + EXTRA.get(src, []) + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    _prune(objs)
    if force or jobs or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


def _prune(objs, keep_variants=False):
    """Objects of A/B variants (build_variant: <src>.<tag>.o, libair_hip.<tag>.so) travel to every GPU lease with the
    snapshot: a default build removes them unless AIR_KEEP_VARIANTS=1."""
    if keep_variants or os.environ.get("AIR_KEEP_VARIANTS", "0") == "1":
        return
    want = set(os.path.basename(o) for o in objs)
    for f in os.listdir(OBJ):
        if f.endswith(".o") and f not in want:
            os.remove(os.path.join(OBJ, f))
    for f in os.listdir(LIBDIR):
        if f.startswith("libair_hip.") and f.endswith(".so") and f != os.path.basename(LIB):
            os.remove(os.path.join(LIBDIR, f))


def build_variant(tag, src, defines):
    """A/B tooling: libair_hip.<tag>.so = the current objects with `src` recompiled under extra -D flags
    (``python -m asvspoof2021_air_amd.build --variant v1 conv_wino4.hip -DW4_X=1``); load it with
    AIR_HIP_LIB=<path> (see _hip.py).  The default library is untouched."""
    os.environ["AIR_KEEP_VARIANTS"] = "1"  # an A/B session holds several variants side by side
    build(verbose=False)
    s = os.path.join(CSRC, src)
    o = os.path.join(OBJ, "%s.%s.o" % (src[:-4], tag))
    r = subprocess.run([HIPCC] + CFLAGS + EXTRA.get(src, []) + list(defines) + ["-c", s, "-o", o], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr)
    objs = [os.path.join(OBJ, f[:-4] + ".o") for f in _sources() if f != src] + [o]
    lib = os.path.join(LIBDIR, "libair_hip.%s.so" % tag)
    r = subprocess.run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr)
    return lib


if __name__ == "__main__":
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2], sys.argv[i + 3:]))
    else:
        print(build(force="--force" in sys.argv))
