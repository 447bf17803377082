"""Build libfar3d_hip.so (every HIP kernel + the C-ABI) for gfx950, in-tree.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box with the
repo snapshot.  Objects are cached per source by mtime so iterating on one kernel is cheap.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libfar3d_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
         "-Wno-unused-result", "-I", os.path.join(HERE, "..", "include")]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _newest_header():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    hs += [os.path.join(HERE, "..", "include", f) for f in os.listdir(os.path.join(HERE, "..", "include"))]
    return max(os.path.getmtime(h) for h in hs) if hs else 0.0


def _compile(src, objdir=None, extra=()):
    obj = os.path.join(objdir or OBJ, src[:-4] + ".o")
    sp = os.path.join(CSRC, src)
    if os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(sp), _newest_header()):
        return obj, False
    cmd = [HIPCC] + FLAGS + list(extra) + ["-c", sp, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    return obj, True


def _stamp_commit():
    """Record the commit this tree was built from (far3d_amd/_build_commit.txt, git-ignored like the .so): the GPU box receives a
    snapshot without .git, and bench.py reports the commit in its JSON line."""
    root = os.path.join(HERE, "..")
    if not os.path.isdir(os.path.join(root, ".git")):
        return
    try:
        c = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip()
        d = subprocess.run(["git", "-C", root, "status", "--porcelain", "--untracked-files=no"], capture_output=True, text=True, timeout=10).stdout.strip()
        if c:
            with open(os.path.join(HERE, "_build_commit.txt"), "w") as f:
                f.write(c + ("+dirty" if d else "") + "\n")
    except Exception:   # noqa: BLE001  (never fail a build over the stamp)
        pass


def build(force=False, verbose=True):
    _stamp_commit()
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        res = list(ex.map(_compile, srcs))
    objs = [o for o, _ in res]
    if any(c for _, c in res) or not os.path.exists(LIB) or \
            os.path.getmtime(LIB) < max(os.path.getmtime(o) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("[far3d_amd.build] linked", LIB, "(%d objects)" % len(objs))
    elif verbose:
        print("[far3d_amd.build] up to date:", LIB)
    return LIB


PROF_LIB = os.path.join(HERE, "libfar3d_hip_prof.so")


def build_profiling(verbose=True):
    """libfar3d_hip_prof.so: the same sources with -DFAR3D_PROFILING (s_memtime stamps at phase boundaries; tools/ only, never the
    product library).  Objects cached per source under csrc/_obj_prof, compiled in parallel like build()."""
    objdir = os.path.join(CSRC, "_obj_prof")
    os.makedirs(objdir, exist_ok=True)
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        res = list(ex.map(lambda f: _compile(f, objdir, ("-DFAR3D_PROFILING",)), srcs))
    objs = [o for o, _ in res]
    if any(c for _, c in res) or not os.path.exists(PROF_LIB) or os.path.getmtime(PROF_LIB) < max(os.path.getmtime(o) for o in objs):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", PROF_LIB] + objs, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("[far3d_amd.build] linked", PROF_LIB)
    return PROF_LIB


if __name__ == "__main__":
    if "--profiling" in sys.argv:
        build_profiling()
        sys.exit(0)
    build(force="--force" in sys.argv)
