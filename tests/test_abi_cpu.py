"""CPU-side checks of the boundary: the library loads and exports every symbol the header
declares, the host-side packing realises the identity the kernel relies on, and the Python
host logic mirrors the reference.  No compute call needs a GPU here."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, synthetic_case
from oracle import gp_oracle

from gp_emulator_amd import GaussianProcess, _gpu_predict, _lib


def header_functions():
    text = open(os.path.join(ROOT, "include", "gp_predict_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gp_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libgp_predict_hip.so does not export %s" % n
        assert n in _lib.SIGNATURES, "no ctypes signature for %s" % n
    assert set(_lib.SIGNATURES) == set(names)
    assert b"gfx950" in lib.gp_version_string()


def test_library_on_disk_is_built_from_these_sources():
    """The in-tree .so travels to the GPU box as it is: it must be the build of the sources beside it
    (build.py stamps the library with a digest of csrc/, include/ and the flags)."""
    from gp_emulator_amd import build as gpbuild
    stamp = gpbuild.LIB + ".digest"
    assert os.path.exists(gpbuild.LIB) and os.path.exists(stamp), "run python -m gp_emulator_amd.build"
    assert open(stamp).read().strip() == gpbuild.source_digest() + "" + "None", \
        "libgp_predict_hip.so is stale (sources changed, or the last build failed): python -m gp_emulator_amd.build"


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """The boundary is a C ABI: include/gp_predict_hip.h compiles as strict C99 and a C program
    links against the library and calls it (device count, error string, version) -- no C++, no
    Python in between.  What a cgo / JNI / ctypes binding relies on."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "cabi.c"
    src.write_text(
        '#include <stdio.h>\n#include <string.h>\n#include "gp_predict_hip.h"\n'
        "int main(void) {\n"
        "  int n = -1; gp_ctx* ctx = NULL;\n"
        "  int rc = gp_device_count(&n);\n"
        '  if (strstr(gp_version_string(), "gfx950") == NULL) return 2;\n'
        "  if (rc == GP_OK && n > 0) return 0;               /* a GPU box: nothing more to check here */\n"
        "  rc = gp_ctx_create(0, &ctx);                        /* no GPU: must fail, loudly */\n"
        "  if (rc == GP_OK || ctx != NULL) return 3;\n"
        "  if (strlen(gp_last_error_string()) == 0) return 4;\n"
        '  printf("%d %s\\n", rc, gp_last_error_string());\n'
        "  return 0;\n}\n")
    exe = tmp_path / "cabi"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                    str(src), "-o", str(exe), "-L", libdir, "-l:" + os.path.basename(_lib.LIB_PATH),
                    "-Wl,-rpath," + libdir], check=True, timeout=120)
    r = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0, r.stdout


@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_host_thread_pool_under_sanitizers(tmp_path, sanitizer):
    """The helper-thread pool of the host-pointer path (csrc/gp_host_pool.hpp) is plain C++: built here
    with ThreadSanitizer and with ASan + UBSan and stressed (3000 jobs of 1-24 tasks, run_on_worker in
    between, shutdown): every task exactly once, results visible to the caller, no report."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = tmp_path / "host_pool_check"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=" + sanitizer, "-fno-sanitize-recover=all", "-pthread",
           "-I", os.path.join(ROOT, "gp_emulator_amd", "csrc"), os.path.join(ROOT, "tests", "csrc", "host_pool_check.cpp"),
           "-o", str(exe)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    if r.returncode != 0 and ("cannot find" in r.stdout or "unrecognized" in r.stdout):
        pytest.skip("this toolchain has no %s sanitizer runtime" % sanitizer)
    assert r.returncode == 0, r.stdout
    for threads in ("6", "2", "1"):
        r = subprocess.run([str(exe), threads, "3000" if threads == "6" else "500"], stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, timeout=300)
        assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout[-2000:]


def test_no_gpu_fails_loudly_not_silently():
    """On a box without a GPU is_gpu=True must raise, never compute on the CPU."""
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    g = synthetic_case("odd_n37_d3")
    gp = GaussianProcess(g["inputs"], [])
    gp.theta, gp.invQ, gp.invQt = g["theta"], g["invQ"], g["invQt"]
    with pytest.raises(_lib.GpuPredictUnavailable):
        gp.predict(g["testing"], is_gpu=True)
    with pytest.raises(_lib.GpuPredictUnavailable):
        gp.gpu_model()


def test_default_device_is_a_process_setting(monkeypatch):
    """The reference computes on device 0 and has no way to choose; one process per GPU sets
    the device once (or exports GP_DEVICE) and every drop-in entry point follows."""
    import gp_emulator_amd
    monkeypatch.setattr(_lib, "_default_device", [None])
    monkeypatch.setenv("GP_DEVICE", "5")
    assert _lib.default_device() == 5
    gp_emulator_amd.set_default_device(2)
    assert _lib.default_device() == 2
    if _lib.device_count() == 0:
        with pytest.raises(_lib.GpuPredictUnavailable):
            _lib.default_context()


def mfma_rows(dtype):
    """own_sub(r, g): row inside a 16-block of C/D register r for lane group g."""
    if np.dtype(dtype) == np.float64:
        return lambda r, g: 4 * r + g
    return lambda r, g: 4 * g + r


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("name", ["odd_n37_d3", "c1_n100_d5", "c2_n250_d11", "c4_n300_d11"])
def test_packed_fragments_reproduce_quadratic_form(name, dtype):
    """Emulate, in numpy, exactly what the kernel does with the packed buffers:
    16x16x4 matrix-core steps over block pairs I >= J, operands taken lane by lane from
    the fragment buffer, only the k-steps the chosen kernel issues (N = 250 -> 63 of 64,
    N = 300 -> 75 of 76), result rows read back per lane -- and check
    k^T invQ k (invQ NON-symmetric) and mean/gradient against the oracle."""
    g = synthetic_case(name)
    N, D = g["inputs"].shape
    pk = _lib.pack_model(np.exp(g["theta"]), g["inputs"], g["invQt"], g["invQ"], dtype)
    kd, nb, nk = pk["kernel_d"], pk["kernel_nb"], pk["kernel_nk"]
    assert kd >= D and 4 * nk >= N and nb == (nk + 3) // 4
    if name == "c2_n250_d11":
        assert nk == 63
    if name == "c4_n300_d11":
        assert nk == 75
    ds = (kd + 2 + 3) & ~3
    xa = pk["xa"].astype(np.float64).reshape(16 * nb, ds)
    frags = pk["frags"].astype(np.float64).reshape(-1, 64)
    sd = pk["sd"].astype(np.float64)
    ctr = pk["centre"].astype(np.float64)
    own = mfma_rows(dtype)
    lanes = np.arange(64)
    grp, col = lanes >> 4, lanes & 15
    # every training point sits in exactly one slot; slots of k-steps >= nk are zero padding
    live = np.zeros(16 * nb, bool)
    for q in range(nk):
        for gq in range(4):
            live[16 * (q // 4) + own(q % 4, gq)] = True
    assert int(np.count_nonzero(np.any(xa != 0, axis=1))) == N
    assert not np.any(xa[~live])

    T = g["testing"][:16]
    tp = np.zeros((16, kd))
    tp[:, :D] = sd[:D] * (T.astype(dtype).astype(np.float64) - ctr[:D])
    # K tile, mean, gradient the way phase A forms them (live k-steps only):
    #   k_i = exp(h_i + g + x''_i . t''),  G_d = sum_i w_i x''_id,  grad = sd (G - t'' mu)
    gm = -0.5 * np.sum(tp * tp, axis=1)
    K = np.exp(xa[:, kd + 1][:, None] + gm[None, :] + xa[:, :kd] @ tp.T)   # (NP, 16)
    K[~live] = np.nan                                      # never computed by the kernel
    w = K[live] * xa[live, kd][:, None]
    mu = w.sum(axis=0)
    G = w.T @ xa[live, :kd]
    grad = ((G - tp * mu[:, None]) * sd[None, :])[:, :D]
    # phase B
    quad = np.zeros(16)
    for J in range(nb):
        acc = np.zeros((16, 16))                              # [row j_local][col m]
        for I in range(J, nb):
            for s in range(4):
                if 4 * I + s >= nk:
                    continue
                f = frags[_lib.load().gp_frag_index(nb, I, J, s)]
                A = np.zeros((16, 4))
                A[col, grp] = f                               # A[row = l&15][k = l>>4]
                i_of_k = np.array([16 * I + own(s, gq) for gq in range(4)])
                B = K[i_of_k, :]                              # B[k][col = m]
                acc += A @ B
        for gq in range(4):
            for r in range(4):
                if 4 * J + r < nk:
                    quad += acc[own(r, gq), :] * K[16 * J + own(r, gq), :]
    var = pk["b"] - quad
    o_mu, o_var, o_der = gp_oracle.cpu_predict(g["inputs"], g["theta"], g["invQ"], g["invQt"], T)
    tol = 1e-12 if np.dtype(dtype) == np.float64 else 5e-6
    assert gp_oracle.maxnorm_err(o_mu, mu) < tol
    assert gp_oracle.maxnorm_err(o_var, var) < tol
    assert gp_oracle.maxnorm_err(o_der, grad) < tol


def test_pack_kernel_choice_and_unsupported_shapes():
    """N <= 320 and D <= 16: fused MFMA kernel; up to N = 1024, D = 64: general-shape kernel
    (kernel_nb == 0, invQ kept as given); beyond: refused."""
    pk = _lib.pack_model(np.ones(13), np.zeros((250, 11)), np.zeros(250), np.zeros((250, 250)))
    assert (pk["kernel_d"], pk["kernel_nb"]) == (11, 16)
    pk = _lib.pack_model(np.ones(12), np.zeros((300, 10)), np.zeros(300), np.zeros((300, 300)))
    assert (pk["kernel_d"], pk["kernel_nb"]) == (10, 19)
    rs = np.random.RandomState(3)
    q = rs.rand(400, 400)
    pk = _lib.pack_model(np.ones(6), rs.rand(400, 4), np.zeros(400), q)
    assert pk["kernel_nb"] == 0 and pk["kernel_d"] == 4
    assert np.array_equal(pk["frags"].reshape(400, 400), q)
    pk = _lib.pack_model(np.ones(40), np.zeros((10, 38)), np.zeros(10), np.zeros((10, 10)))
    assert pk["kernel_nb"] == 0 and pk["kernel_d"] == 38
    with pytest.raises(_lib.GpuPredictError):
        _lib.pack_model(np.ones(72), np.zeros((10, 70)), np.zeros(10), np.zeros((10, 10)))
    with pytest.raises(_lib.GpuPredictError):
        _lib.pack_model(np.ones(4), np.zeros((1100, 2)), np.zeros(1100), np.zeros((1100, 1100)))


BLOCK_CASES = [(250000, 100000), (200000, 100000), (1000, 200000), (200001, 100000),
               (300001, 100000), (7, 3), (1, 5), (100000, 1e5), (900000, 1e5)]


@pytest.mark.parametrize("args", BLOCK_CASES)
def test_get_gpu_block_matches_oracle(args):
    gp = GaussianProcess(np.zeros((3, 2)), [])
    s, e = gp.get_gpu_block(*args)
    os_, oe = gp_oracle.get_gpu_block(*args)
    assert list(s) == list(os_) and list(e) == list(oe)


def test_host_numpy_branch_and_hessian_match_oracle():
    g = synthetic_case("odd_n37_d3")
    gp = GaussianProcess(g["inputs"], [])
    gp.theta, gp.invQ, gp.invQt = g["theta"], g["invQ"], g["invQt"]
    mu, var, der = gp.predict(g["testing"])
    assert gp_oracle.maxnorm_err(g["mu"], mu) < 1e-13
    assert gp_oracle.maxnorm_err(g["var"], var) < 1e-13
    assert gp_oracle.maxnorm_err(g["deriv"], der) < 1e-13
    mu2, der2 = gp.predict(g["testing"], do_unc=False)
    assert np.array_equal(mu, mu2)
    h = gp.hessian(g["testing"][:33])
    assert gp_oracle.maxnorm_err(g["hess"], h) < 1e-12


def test_set_params_matches_oracle():
    from conftest import load_golden
    g = load_golden("prosail_pc0")
    gp = GaussianProcess(g["inputs"], g["targets"])
    gp._set_params(g["theta"])
    pl = gp_oracle.prepare_likelihood(g["inputs"], g["targets"], g["theta"])
    assert np.max(np.abs(gp.invQ - pl["invQ"])) / np.max(np.abs(pl["invQ"])) < 1e-6
    assert np.max(np.abs(gp.invQt - pl["invQt"])) / np.max(np.abs(pl["invQt"])) < 1e-6
    assert abs(gp.logdetQ - pl["logdetQ"]) < 1e-6 * abs(pl["logdetQ"])


def test_predict_wrap_argument_checks():
    """The reference exit()s on a wrong dtype/ndim (_gpu_predict.cpp:39-58); the shim raises."""
    f = np.zeros(4, np.float64)
    good = [np.ones(3), np.zeros(2), np.zeros(2), np.zeros(4), np.zeros(2), np.zeros(2),
            np.zeros(2), np.zeros(2)]
    bad = list(good)
    bad[4] = np.zeros(2, np.float32)
    with pytest.raises(TypeError):
        _gpu_predict.predict_wrap(*bad, 2, 2, 1, 3)
    bad = list(good)
    bad[1] = np.zeros((2, 1))
    with pytest.raises(ValueError):
        _gpu_predict.predict_wrap(*bad, 2, 2, 1, 3)
    bad = list(good)
    bad[0] = np.ones(3, np.int64)
    with pytest.raises(TypeError):
        _gpu_predict.predict_wrap(*[a.astype(np.int64) for a in good], 2, 2, 1, 3)
    with pytest.raises(ValueError):
        _gpu_predict.predict_wrap(*good, 5, 2, 1, 3)   # outputs too short for n_predict
    with pytest.raises(TypeError):
        _gpu_predict.predict_wrap([1.0, 2, 3], *good[1:], 2, 2, 1, 3)
    del f


def make_mv_dump(tmp_path):
    """Write the golden emulator back out in the reference's .npz layout."""
    from conftest import load_golden
    g = load_golden("prosail_mv")
    # X is only used through compress(X) = train_data; rebuild an X with that projection
    X = g["train_data"].T @ g["basis_functions"]
    path = tmp_path / "emu.npz"
    np.savez_compressed(path, X=X, y=g["y_train"], hyperparams=g["hyperparams"], thresh=0.99,
                        basis_functions=g["basis_functions"], n_pcs=int(g["n_pcs"]))
    return g, str(path)


def test_multivariate_emulator_cpu_matches_reference_outputs(tmp_path):
    """MultivariateEmulator(dump=...).predict(y) against the reference's own outputs
    (tests/golden/prosail_mv.npz), numpy branch."""
    from gp_emulator_amd import MultivariateEmulator
    g, path = make_mv_dump(tmp_path)
    mv = MultivariateEmulator(dump=path)
    assert mv.n_pcs == 12 and len(mv.emulators) == 12
    # the basis rows are orthonormal, so compress(X) reproduces train_data
    assert np.max(np.abs(mv.compress(mv.X_train) - g["train_data"])) < 1e-9
    for j, p in enumerate(g["points"][:2]):
        fwd, jac = mv.predict(p)
        assert fwd.shape == (2101,) and jac.shape == (10, 2101)
        assert np.max(np.abs(fwd - g["fwd"][j])) <= 1e-6
        assert np.max(np.abs(jac - g["jac"][j])) / np.max(np.abs(g["jac"][j])) <= 1e-5
    assert np.max(np.abs(mv.predict(g["points"][0], do_deriv=False) - g["fwd"][0])) <= 1e-6
    many = mv.predict_many(g["points"], is_gpu=False)
    assert np.max(np.abs(many - g["fwd"])) <= 1e-6
    out = tmp_path / "again.npz"
    mv.dump_emulator(str(out))
    with np.load(str(out), allow_pickle=False) as f:
        assert sorted(f.files) == ["X", "basis_functions", "hyperparams", "n_pcs", "thresh", "y"]


def test_host_training_objective_numpy_branch():
    from conftest import load_golden
    g = load_golden("training_objective")
    gp = GaussianProcess(g["smooth_inputs"], g["smooth_targets"])
    for k in range(3):
        ll = gp.loglikelihood(g["smooth_thetas"][k])
        gr = gp.partial_devs(g["smooth_thetas"][k])
        assert abs(ll - g["smooth_loglik"][k]) <= 1e-9 * abs(g["smooth_loglik"][k])
        assert np.max(np.abs(gr - g["smooth_grad"][k])) <= 1e-7 * np.max(np.abs(g["smooth_grad"][k]))
    # noisy targets, so that the optimum is interior (with the noiseless ones the reference's
    # own optimiser drives the noise term to 0 until Cholesky fails and reports cost 9999)
    noisy = g["smooth_targets"] + 0.05 * np.random.RandomState(1).standard_normal(120)
    gp = GaussianProcess(g["smooth_inputs"], noisy)
    np.random.seed(1)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cost, theta = gp.learn_hyperparameters(n_tries=2)
    assert theta.shape == (6,) and abs(cost - (-148.607048555)) < 1e-5
    assert np.max(np.abs(gp.partial_devs(theta))) < 1e-3
    mu, var, der = gp.predict(g["smooth_inputs"][:5])
    assert np.max(np.abs(mu - g["smooth_targets"][:5])) < 0.1


def test_content_digest_sees_every_byte_and_host_blocks_follow_edits():
    """gp_content_digest (no GPU): the digest the device-resident copies are keyed by.  Any single-byte change in
    any block changes it, block boundaries and lengths count, unaligned and odd-length blocks work; HostBlocks
    re-reads strided members and lists on every digest."""
    import ctypes
    lib = _lib.load()
    rs = np.random.RandomState(0)

    def digest(arrs):
        ptrs = (ctypes.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        lens = (ctypes.c_int64 * len(arrs))(*[a.nbytes for a in arrs])
        return int(lib.gp_content_digest(ptrs, lens, len(arrs)))
    blocks = [rs.random_sample(n) for n in (1, 7, 31, 32, 33, 250 * 250, 13)] + [rs.bytes(37), rs.bytes(8)]
    blocks = [np.frombuffer(b, np.uint8).copy() if isinstance(b, bytes) else b for b in blocks]
    d0 = digest(blocks)
    assert d0 == digest([b.copy() for b in blocks]) and d0 != 0
    for bi, b in enumerate(blocks):                      # one byte of every block, first / middle / last
        raw = b.view(np.uint8)
        for pos in {0, raw.size // 2, raw.size - 1}:
            raw[pos] ^= 1
            assert digest(blocks) != d0, (bi, pos)
            raw[pos] ^= 1
    assert digest(blocks) == d0
    assert digest(blocks[::-1]) != d0                    # order counts
    assert digest(blocks[:-1] + [blocks[-1][:4], blocks[-1][4:]]) != d0      # so do the block boundaries
    odd = np.frombuffer(rs.bytes(8 * 40 + 3), np.uint8)[3:]                    # an unaligned block
    assert digest([odd]) == digest([odd.copy()])
    # HostBlocks: contiguous arrays by pointer, a strided column and a list through re-filled buffers
    H = rs.random_sample((12, 13))
    a, lst = rs.random_sample((50, 50)), [1.0, 2.0, 3.0]
    hb = _lib.HostBlocks([a, H[:, 3], lst])
    assert hb.unchanged() and hb.same_arrays([a, H[:, 3], lst]) is False      # (a fresh slice is a new object)
    col = H[:, 3]
    hb = _lib.HostBlocks([a, col, lst])
    assert hb.same_arrays([a, col, lst]) and hb.unchanged()
    a[17, 4] += 1e-12
    assert not hb.unchanged()
    a[17, 4] -= 1e-12
    H[5, 3] *= 2.0
    assert not hb.unchanged()
    H[5, 3] /= 2.0
    assert hb.unchanged()
    lst[2] = 4.0
    assert not hb.unchanged()
