"""Build libgp_predict_hip.so for gfx950 with plain hipcc (no cmake, no torch extension).

    python -m gp_emulator_amd.build [--force] [--jobs N]

One object per (compute dtype, NB) kernel translation unit + the C-ABI object, linked into
``gp_emulator_amd/libgp_predict_hip.so`` (in-tree, git-ignored, shipped to the GPU box by
gpurun).  hipcc cross-compiles without a GPU.  Replaces the reference's CMake/CUDA build
(CMakeLists.txt, gp_emulator/gpu/CMakeLists.txt, setup.py:16-35), which picks ONE precision
at configure time; here both precisions are in the one library.
"""
import argparse
import concurrent.futures
import hashlib
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libgp_predict_hip.so")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast",
         "-I" + CSRC, "-I" + os.path.join(ROOT, "include")]


def kernel_nbs():
    text = open(os.path.join(CSRC, "gp_dispatch.hpp")).read()
    m = re.search(r"GP_FOR_EACH_KERNEL_NB\(X\)(.*)", text)
    return [int(x) for x in re.findall(r"X\((\d+)\)", m.group(1))]


def source_digest():
    h = hashlib.sha256()
    for d in (CSRC, os.path.join(ROOT, "include")):
        for name in sorted(os.listdir(d)):
            p = os.path.join(d, name)
            if os.path.isfile(p) and name.endswith((".hip", ".hpp", ".h")):
                h.update(name.encode())
                h.update(open(p, "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def build(force=False, jobs=None, verbose=True, defines=(), lib=None, only_nb=None):
    """Compile and link.  ``defines`` (e.g. ["-DGP_WAVES=4"]), ``lib`` (output path) and
    ``only_nb`` (restrict the kernel set) exist for tools/ab_bench.py variants only."""
    global OBJ, LIB
    variant = bool(defines or lib or only_nb)
    obj_dir, lib_path = OBJ, LIB
    if variant:
        tag = hashlib.sha256((" ".join(defines) + str(only_nb)).encode()).hexdigest()[:10]
        obj_dir = os.path.join(CSRC, "_obj", "variant_" + tag)
        lib_path = lib or os.path.join(HERE, "libgp_predict_hip_%s.so" % tag)
    return _build(force, jobs, verbose, list(defines), obj_dir, lib_path, only_nb)


def _build(force, jobs, verbose, defines, OBJ, LIB, only_nb):
    os.makedirs(OBJ, exist_ok=True)
    stamp = LIB + ".digest"      # travels with the .so (the object dir does not)
    digest = source_digest() + " ".join(defines) + str(only_nb)
    if (not force and os.path.exists(LIB) and os.path.exists(stamp)
            and open(stamp).read().strip() == digest):
        if verbose:
            print("[gp build] up to date:", LIB)
        return LIB
    jobs = jobs or min(8, os.cpu_count() or 1)
    tasks = []
    for tname, ctype in (("f64", "double"), ("f32", "float")):
        for nb in kernel_nbs():
            obj = os.path.join(OBJ, "kern_%s_%d.o" % (tname, nb))
            tasks.append([HIPCC] + FLAGS + defines + ["-DGP_T=" + ctype, "-DGP_TNAME=" + tname,
                                            "-DGP_NB=%d" % nb, "-c",
                                            os.path.join(CSRC, "gp_kernels_tu.hip"), "-o", obj])
            obj = os.path.join(OBJ, "hessm_%s_%d.o" % (tname, nb))
            tasks.append([HIPCC] + FLAGS + defines + ["-DGP_T=" + ctype, "-DGP_TNAME=" + tname,
                                            "-DGP_NB=%d" % nb, "-c",
                                            os.path.join(CSRC, "gp_hessian_mfma_tu.hip"), "-o", obj])
        obj = os.path.join(OBJ, "generic_%s.o" % tname)
        tasks.append([HIPCC] + FLAGS + defines + ["-DGP_T=" + ctype, "-DGP_TNAME=" + tname, "-c",
                                        os.path.join(CSRC, "gp_generic_tu.hip"), "-o", obj])
        obj = os.path.join(OBJ, "hess_%s.o" % tname)
        tasks.append([HIPCC] + FLAGS + defines + ["-DGP_T=" + ctype, "-DGP_TNAME=" + tname, "-c",
                                        os.path.join(CSRC, "gp_hessian_tu.hip"), "-o", obj])
    tasks.append([HIPCC] + FLAGS + defines + ["-c", os.path.join(CSRC, "gp_reconstruct_tu.hip"),
                                           "-o", os.path.join(OBJ, "reconstruct.o")])
    tasks.append([HIPCC] + FLAGS + defines + ["-c", os.path.join(CSRC, "gp_train_tu.hip"),
                                           "-o", os.path.join(OBJ, "train.o")])
    abi_obj = os.path.join(OBJ, "gp_abi.o")
    tasks.append([HIPCC] + FLAGS + defines + ["-c", os.path.join(CSRC, "gp_abi.hip"), "-o", abi_obj])
    # biggest kernels first so the pool drains evenly
    tasks.sort(key=lambda c: -int(next((a[8:] for a in c if a.startswith("-DGP_NB=")), "0")))
    if verbose:
        print("[gp build] compiling %d translation units with %d jobs" % (len(tasks), jobs))
    with concurrent.futures.ThreadPoolExecutor(jobs) as ex:
        for out in ex.map(run, tasks):
            if out.strip() and verbose:
                print(out)
    objs = [c[-1] for c in tasks]
    run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs)
    with open(stamp, "w") as fh:
        fh.write(digest)
    if verbose:
        print("[gp build] wrote", LIB)
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=None)
    ap.add_argument("--define", action="append", default=[], help="extra -D flag (variant build)")
    ap.add_argument("--lib", default=None, help="output path (variant build)")
    a = ap.parse_args()
    try:
        build(force=a.force, jobs=a.jobs, defines=["-D" + d for d in a.define], lib=a.lib)
    except RuntimeError as e:
        print(e, file=sys.stderr)
        sys.exit(1)
