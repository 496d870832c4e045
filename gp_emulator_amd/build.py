"""Build libgp_predict_hip.so for gfx950 with plain hipcc (no cmake, no torch extension).

    python -m gp_emulator_amd.build [--force] [--jobs N]

One object per (compute dtype, kernel size) translation unit + the C-ABI object, linked into
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


def kernel_sizes(which):
    """The NK (predict kernels) or NB (matrix-core Hessian kernels) list of gp_dispatch.hpp."""
    text = open(os.path.join(CSRC, "gp_dispatch.hpp")).read()
    m = re.search(r"GP_FOR_EACH_KERNEL_%s\(X\)(.*)" % which, text)
    return [int(x) for x in re.findall(r"X\((\d+)\)", m.group(1))]


def source_digest():
    h = hashlib.sha256()
    for d in (CSRC, os.path.join(ROOT, "include")):
        for name in sorted(os.listdir(d)):
            p = os.path.join(d, name)
            if os.path.isfile(p) and name.endswith((".hip", ".hpp", ".h", ".cpp")):
                h.update(name.encode())
                h.update(open(p, "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _includes(path, seen):
    """Files ``path`` includes with #include "...", transitively (csrc/ and include/ only)."""
    if path in seen or not os.path.exists(path):
        return seen
    seen.add(path)
    for name in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(path).read(), flags=re.M):
        for d in (CSRC, os.path.join(ROOT, "include")):
            _includes(os.path.join(d, name), seen)
    return seen


def unit_digest(cmd):
    """Digest of one compile command: its flags plus the source and every header it pulls in."""
    src = next(a for a in cmd if a.endswith((".hip", ".cpp")))
    h = hashlib.sha256(" ".join(cmd).encode())
    for f in sorted(_includes(src, set())):
        h.update(f.encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def run(cmd):
    """Compile one translation unit unless its object is up to date (digest beside the object)."""
    obj = cmd[-1]
    stamp = obj + ".digest" if "-c" in cmd else None     # (the link step always runs)
    digest = unit_digest(cmd) if stamp else None
    if stamp and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == digest:
        return ""
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), r.stdout))
    if stamp:
        with open(stamp, "w") as fh:
            fh.write(digest)
    return r.stdout


def build(force=False, jobs=None, verbose=True, defines=(), lib=None, only_units=None):
    """Compile and link.  ``defines`` (e.g. ["-DGP_WAVES=4"]), ``lib`` (output path) and
    ``only_units`` exist for tools/ab_bench.py variants only: a variant build compiles into its own
    object directory; with ``only_units`` (object base names, e.g. ["hessm_f64_19"]) just those
    units are compiled with the defines and every other object is taken from the main build."""
    global OBJ, LIB
    variant = bool(defines or lib or only_units)
    obj_dir, lib_path = OBJ, LIB
    if variant:
        tag = hashlib.sha256((" ".join(defines) + str(only_units)).encode()).hexdigest()[:10]
        obj_dir = os.path.join(CSRC, "_obj", "variant_" + tag)
        lib_path = lib or os.path.join(HERE, "libgp_predict_hip_%s.so" % tag)
    return _build(force, jobs, verbose, list(defines), obj_dir, lib_path, only_units)


def _build(force, jobs, verbose, defines, OBJ, LIB, only_units):
    main_obj = os.path.join(CSRC, "_obj")
    os.makedirs(OBJ, exist_ok=True)
    stamp = LIB + ".digest"      # travels with the .so (the object dir does not)
    digest = source_digest() + " ".join(defines) + str(only_units)
    if (not force and os.path.exists(LIB) and os.path.exists(stamp)
            and open(stamp).read().strip() == digest):
        if verbose:
            print("[gp build] up to date:", LIB)
        return LIB
    if force:                      # forget every per-unit digest: recompile everything
        for name in os.listdir(OBJ):
            if name.endswith(".o.digest"):
                os.remove(os.path.join(OBJ, name))
    jobs = jobs or min(8, os.cpu_count() or 1)
    tasks = []
    for tname, ctype in (("f64", "double"), ("f32", "float")):
        for nk in kernel_sizes("NK"):
            obj = os.path.join(OBJ, "kern_%s_%d.o" % (tname, nk))
            tasks.append([HIPCC] + FLAGS + defines + ["-DGP_T=" + ctype, "-DGP_TNAME=" + tname,
                                            "-DGP_NK=%d" % nk, "-c",
                                            os.path.join(CSRC, "gp_kernels_tu.hip"), "-o", obj])
        for nb in kernel_sizes("NB"):
            obj = os.path.join(OBJ, "hessm_%s_%d.o" % (tname, nb))
            tasks.append([HIPCC] + FLAGS + defines + ["-DGP_T=" + ctype, "-DGP_TNAME=" + tname,
                                            "-DGP_NB=%d" % nb, "-c",
                                            os.path.join(CSRC, "gp_hessian_mfma_tu.hip"), "-o", obj])
        obj = os.path.join(OBJ, "generic_%s.o" % tname)
        tasks.append([HIPCC] + FLAGS + defines + ["-DGP_T=" + ctype, "-DGP_TNAME=" + tname, "-c",
                                        os.path.join(CSRC, "gp_generic_tu.hip"), "-o", obj])
        obj = os.path.join(OBJ, "few_%s.o" % tname)
        tasks.append([HIPCC] + FLAGS + defines + ["-DGP_T=" + ctype, "-DGP_TNAME=" + tname, "-c",
                                        os.path.join(CSRC, "gp_few_tu.hip"), "-o", obj])
        obj = os.path.join(OBJ, "hess_%s.o" % tname)
        tasks.append([HIPCC] + FLAGS + defines + ["-DGP_T=" + ctype, "-DGP_TNAME=" + tname, "-c",
                                        os.path.join(CSRC, "gp_hessian_tu.hip"), "-o", obj])
    tasks.append([HIPCC] + FLAGS + defines + ["-c", os.path.join(CSRC, "gp_reconstruct_tu.hip"),
                                           "-o", os.path.join(OBJ, "reconstruct.o")])
    tasks.append([HIPCC] + FLAGS + defines + ["-c", os.path.join(CSRC, "gp_train_tu.hip"),
                                           "-o", os.path.join(OBJ, "train.o")])
    # plain host C++ (no device pass): the content digest with its per-ISA clones
    tasks.append([HIPCC, "-O3", "-std=c++17", "-fPIC", "-x", "c++", "-c", os.path.join(CSRC, "gp_host_digest.cpp"),
                  "-o", os.path.join(OBJ, "host_digest.o")])
    abi_obj = os.path.join(OBJ, "gp_abi.o")
    tasks.append([HIPCC] + FLAGS + defines + ["-c", os.path.join(CSRC, "gp_abi.hip"), "-o", abi_obj])
    # biggest kernels first so the pool drains evenly
    def weight(c):
        for a in c:
            if a.startswith("-DGP_NK="):
                return int(a[8:])
            if a.startswith("-DGP_NB="):
                return 4 * int(a[8:])
        return 0
    tasks.sort(key=lambda c: -weight(c))
    objs = [c[-1] for c in tasks]
    if only_units:      # variant: the named units only; the rest comes from the main build's objects
        keep = lambda o: os.path.basename(o)[:-2] in only_units
        objs = [o if keep(o) else os.path.join(main_obj, os.path.basename(o)) for o in objs]
        tasks = [c for c in tasks if keep(c[-1])]
        missing = [o for o in objs if not keep(o) and not os.path.exists(o)]
        if missing:
            raise RuntimeError("variant build needs the main build's objects first: " + missing[0])
    if verbose:
        print("[gp build] compiling %d translation units with %d jobs" % (len(tasks), jobs))
    with concurrent.futures.ThreadPoolExecutor(jobs) as ex:
        for out in ex.map(run, tasks):
            if out.strip() and verbose:
                print(out)
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
    ap.add_argument("--only", action="append", default=[],
                    help="variant build: compile only this unit (object base name, e.g. hessm_f64_19); "
                         "the other objects come from the main build")
    a = ap.parse_args()
    try:
        build(force=a.force, jobs=a.jobs, defines=["-D" + d for d in a.define], lib=a.lib, only_units=a.only or None)
    except RuntimeError as e:
        print(e, file=sys.stderr)
        print("[gp build] FAILED (the library on disk, if any, is from an earlier build)", flush=True)   # last line of both streams
        print("[gp build] FAILED", file=sys.stderr)
        sys.exit(1)
