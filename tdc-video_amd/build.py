"""Build libtdc_hip.so (gfx950) in-tree with hipcc.  Used by __graft_entry__.build() and by lib.load() when the
library is missing (under a file lock).  hipcc cross-compiles without a GPU."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libtdc_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
# attention reads its S / O accumulators with VALU every KV tile: keep MFMA results in VGPRs, otherwise the compiler
# parks them in AGPRs and pays 160 v_accvgpr moves per tile (GEMM accumulators are only read in the epilogue, AGPRs fit)
# -fno-honor-nans: fmaxf without the canonicalising v_max x,x (scores are finite; -inf only enters through the mask)
FILE_FLAGS = {"attention.hip": ["-mllvm", "--amdgpu-mfma-vgpr-form", "-fno-honor-nans"],
              "attention32.hip": ["-mllvm", "--amdgpu-mfma-vgpr-form", "-fno-honor-nans"]}


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip") or f.endswith(".cpp"))


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "tdc_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)

    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + \
           [os.path.join(HERE, "..", "include", "tdc_hip.h"), os.path.abspath(__file__)]
    t_hdr = max(os.path.getmtime(h) for h in hdrs)

    def cc(src):
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        # per-object staleness: a translation unit is recompiled when it, a header or this script is newer than its object
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(t_hdr, os.path.getmtime(os.path.join(CSRC, src))):
            return obj
        cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(src, []) + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr))
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(cc, sources()))
    # link to a temporary name and rename into place: a process that starts while the linker is writing never sees (and
    # dlopens) a partial library
    tmp = OUT + ".tmp.%d" % os.getpid()
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs,
                       capture_output=True, text=True)
    if r.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError("link failed:\n" + r.stderr)
    os.replace(tmp, OUT)
    if verbose:
        print("[tdc-video_amd] built", OUT, file=sys.stderr)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
