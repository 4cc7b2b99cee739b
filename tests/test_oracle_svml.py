"""CPU: the restatement of NumPy's float32 arctan2 / arcsin (oracle/svml_f32.c -- the SVML kernels NumPy dispatches to on
AVX512_SKX machines, VRCP14PS / VRSQRT14PS included) is pinned on NumPy itself: on the committed vectors NumPy produced on the
generating machine (any host), and live on tens of millions of inputs wherever this host's NumPy takes the same dispatch."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import build_oracle as B

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _numpy_runs_svml():
    try:
        from numpy._core._multiarray_umath import __cpu_features__ as feats
    except Exception:
        return False
    return bool(feats.get("AVX512_SKX")) and os.uname().machine == "x86_64"


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _same(a, b):
    """bit-identical, NaN payloads aside"""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return ((_bits(a) == _bits(b)) | (np.isnan(a) & np.isnan(b))).all()


def test_tables_of_oracle_and_product_are_the_same_file():
    a = open(os.path.join(ROOT, "oracle", "approx14_tables.h")).read()
    b = open(os.path.join(ROOT, "overlapnet_amd", "csrc", "approx14_tables.h")).read()
    assert a == b and a.count("static const") == 4


def test_against_the_vectors_numpy_produced():
    with np.load(os.path.join(ROOT, "tests", "golden", "svml_f32_vectors.npz")) as z:
        x, y, at, s, asn = z["x"], z["y"], z["atan2"], z["s"], z["asin"]
    assert len(x) > 90000 and len(s) > 90000
    assert _same(B.svml_arctan2(y, x), at)
    assert _same(B.svml_arcsin(s), asn)
    # the vectors are NOT what a correctly rounded function gives: the restatement is doing real work
    fin = np.isfinite(x) & np.isfinite(y)
    cr = np.arctan2(y[fin].astype(np.float64), x[fin].astype(np.float64)).astype(np.float32)
    assert (_bits(cr) != _bits(at[fin])).mean() > 0.05


@pytest.mark.skipif(not _numpy_runs_svml(), reason="this host's NumPy does not dispatch float32 arctan2 / arcsin to SVML")
def test_against_numpy_itself_on_1e8_inputs():
    rng = np.random.default_rng(7)
    n = 12_500_000
    with np.errstate(all="ignore"):
        for k in range(4):
            if k % 2 == 0:
                x = rng.normal(0, 25, n).astype(np.float32)
                y = rng.normal(0, 25, n).astype(np.float32)
            else:
                x = (rng.uniform(-1, 1, n) * np.exp(rng.uniform(-40, 40, n))).astype(np.float32)
                y = (rng.uniform(-1, 1, n) * np.exp(rng.uniform(-40, 40, n))).astype(np.float32)
            assert np.array_equal(_bits(B.svml_arctan2(y, x)), _bits(np.arctan2(y, x))), k
        for k in range(4):
            s = rng.uniform(-1, 1, n).astype(np.float32) if k % 2 == 0 else \
                (np.sign(rng.uniform(-1, 1, n)) * (1 - np.exp(rng.uniform(-25, 0, n)))).astype(np.float32)
            assert np.array_equal(_bits(B.svml_arcsin(s)), _bits(np.arcsin(s))), k


@pytest.mark.skipif(not (shutil.which("gcc") and "avx512f" in open("/proc/cpuinfo").read()), reason="needs an AVX-512 CPU")
def test_approx14_tables_against_the_instructions(tmp_path):
    """VRCP14PS / VRSQRT14PS executed on this CPU vs the integer tables, over exponents and mantissas."""
    src = tmp_path / "t.c"
    src.write_text(r"""
#include <immintrin.h>
#include <stdio.h>
int main(int argc, char** argv) {
  FILE* f = fopen(argv[1], "rb"); static float in[1 << 20], o1[1 << 20], o2[1 << 20];
  size_t n = fread(in, 4, 1 << 20, f); fclose(f);
  for (size_t i = 0; i < n; i += 16) {
    _mm512_storeu_ps(o1 + i, _mm512_rcp14_ps(_mm512_loadu_ps(in + i)));
    _mm512_storeu_ps(o2 + i, _mm512_rsqrt14_ps(_mm512_loadu_ps(in + i)));
  }
  f = fopen(argv[2], "wb"); fwrite(o1, 4, n, f); fwrite(o2, 4, n, f); fclose(f); return 0; }
""")
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-O2", "-mavx512f", str(src), "-o", str(exe)])
    rng = np.random.default_rng(3)
    n = 1 << 20
    x = (rng.uniform(1, 2, n) * 2.0 ** rng.integers(-100, 100, n)).astype(np.float32)
    x[:64] = 2.0 ** np.arange(-32, 32, dtype=np.float32)            # exact powers of two
    x.tofile(tmp_path / "in.bin")
    subprocess.check_call([str(exe), str(tmp_path / "in.bin"), str(tmp_path / "out.bin")])
    out = np.fromfile(tmp_path / "out.bin", np.float32)
    lib = B.load()
    idx = np.concatenate([np.arange(4096), rng.integers(0, n, 60000)])
    for i in idx:
        assert np.float32(lib.ovn_rcp14(float(x[i]))) == out[i], x[i]
        assert np.float32(lib.ovn_rsqrt14(float(x[i]))) == out[n + i], x[i]
