"""Build libtreelearn_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m treelearn_amd.build [--force] [--dev]

`--dev` also compiles the developer variants (ablation / segment-timer instantiations of the conv kernels, the gather
micro-benchmarks of tl_dev.hip and their `tl_dev_*` hooks, which tools/dev_*.py drive, and the window conv kernel tl_conv_win.hip,
an experiment that reached parity with the default kernels but never beat them); the default (release) library contains none of them.

The library is built IN-TREE (treelearn_amd/lib/) so it travels with the repo snapshot to the
GPU box; it is git-ignored.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libtreelearn_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


DEV_ONLY = ("tl_dev.hip", "tl_conv_win.hip")     # the gather micro-benchmarks; the window conv kernel (parity-green, never faster: DESIGN.md 0.3)


# units compiled a SECOND time with -DTL_F16_BUILD (csrc/tl_half.h): the same kernels with IEEE-half conversions and the f16 MFMA,
# launchers suffixed _f16 -- the float16 inference path (TL_F16)
F16_UNITS = ("tl_conv_direct.hip", "tl_conv_blk.hip", "tl_conv_up.hip", "tl_conv_stream.hip", "tl_conv_streamq.hip", "tl_conv_small.hip", "tl_conv_bf16.hip", "tl_head.hip",
             # the training units (csrc/tl_f16_train.h): weight gradients, BatchNorm train forward / backward, row gather / scatter-add, the heads' small Linears
             "tl_wgrad.hip", "tl_wgrad_dense.hip", "tl_wgrad_rows.hip", "tl_bn.hip", "tl_rows.hip", "tl_linear_small.hip")


def sources(dev=False):
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip") and (dev or f not in DEV_ONLY))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, dev=False):
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj_dev" if dev else "obj")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "treelearn_hip.h"))
    srcs = sources(dev)
    flags = FLAGS + (["-DTL_DEV"] if dev else [])
    stamp = os.path.join(LIBDIR, ".flavour")
    flavour = "dev" if dev else "release"
    relink = not os.path.exists(stamp) or open(stamp).read() != flavour
    objs = [os.path.join(objdir, os.path.basename(s)[:-4] + ".o") for s in srcs]
    jobs = [(s, o, []) for s, o in zip(srcs, objs)]
    for s in srcs:
        if os.path.basename(s) in F16_UNITS:
            o = os.path.join(objdir, os.path.basename(s)[:-4] + "_f16.o")
            objs.append(o); jobs.append((s, o, ["-DTL_F16_BUILD", "-UTL_DEV"]))        # (the developer hooks exist once, in the bf16 objects)

    def compile_one(so):
        s, o, extra = so
        if force or _stale(o, [s] + hdrs):
            cmd = [HIPCC] + flags + extra + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
            return True
        return False

    with ThreadPoolExecutor(max_workers=6) as ex:
        changed = list(ex.map(compile_one, jobs))
    if force or relink or any(changed) or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        open(stamp, "w").write(flavour)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, dev="--dev" in sys.argv))
