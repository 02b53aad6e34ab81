"""Builds libf3dg_hip.so (the C-ABI library of include/f3dg.h) for gfx950 with hipcc, in-tree.

Flags that matter for parity (DESIGN.md "numerics"):
  -ffp-contract=off       hipcc defaults to 'fast' (FMA contraction); the GOF exponent cancels 1e5..1e6 x, so the
                          reference's separate multiply/add roundings are kept
  no -ffast-math, IEEE fp32 divide/sqrt (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt), denormals kept.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libf3dg_hip.so")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-fgpu-rdc" if False else "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]
if os.environ.get("F3DG_LAB"):          # builder-side experiments: diagnostic kernels and A/B variants no caller of the library reaches
    FLAGS.append("-DF3DG_LAB")


# Per-file extras. -fno-slp-vectorize: the SLP vectorizer pairs the float32 multiplies / adds of the per-pixel quadric into
# v_pk_mul_f32 / v_pk_add_f32; on gfx950 those issue at half rate, so nothing is gained and the v_mov shuffles that feed them
# are pure overhead (measured on the compositing kernel: DESIGN.md section 5). Results are bit-identical either way.
EXTRA_FLAGS = {name: os.environ.get("F3DG_EXTRA_" + name.split(".")[0].upper(), default).split()
               for name, default in (("f3dg_render.hip", "-fno-slp-vectorize"), ("f3dg_render4.hip", "-fno-slp-vectorize"), ("f3dg_render5.hip", "-fno-slp-vectorize"), ("f3dg_backward.hip", "-fno-slp-vectorize"), ("f3dg_backward5.hip", "-fno-slp-vectorize"),
                                     ("f3dg_integrate.hip", "-fno-slp-vectorize"), ("f3dg_preprocess.hip", "-fno-slp-vectorize"))}


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


STAMP = os.path.join(CSRC, ".build_flags")      # the flag set the objects in the tree were compiled with


def build(force=False, verbose=False):
    # a tree that holds the objects of the OTHER flag set (a lab build left behind, or the default one when F3DG_LAB is asked for) is rebuilt
    # as a whole: the driver's build() must never pick up a -DF3DG_LAB library because its objects look newer than the sources
    stamp = " ".join(FLAGS) + " | " + " ".join("%s:%s" % (k, " ".join(v)) for k, v in sorted(EXTRA_FLAGS.items()))
    try:
        if open(STAMP).read() != stamp:
            force = True
    except OSError:
        force = True
    srcs = sources()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "f3dg.h"))
    objs = [s[:-4] + ".o" for s in srcs]

    def compile_one(pair):
        src, obj = pair
        if force or _stale(obj, [src] + headers):
            cmd = [HIPCC] + FLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            return True
        return False

    with ThreadPoolExecutor(max_workers=min(4, len(srcs))) as ex:
        rebuilt = list(ex.map(compile_one, zip(srcs, objs)))
    if force or any(rebuilt) or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(stamp)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
