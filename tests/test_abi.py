"""The C-ABI library loads and exports every symbol include/*.h declares; with no usable GPU it refuses to
compute (there is no CPU fallback) and never touches the oracle."""
import ctypes as C
import glob
import importlib
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
binding = importlib.import_module("hc-mvs_amd.binding")


def declared_functions():
    names = []
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        src = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names += re.findall(r"^\s*(?:const\s+)?(?:int|void|char\s*\*|const char\s*\*)\s*\**\s*(hcmvs_\w+)\s*\(", src, flags=re.M)
    return sorted(set(names))


def test_every_declared_symbol_is_exported_and_bound():
    names = declared_functions()
    assert len(names) >= 14 and "hcmvs_estimate_device" in names
    lib = C.CDLL(binding.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "libhcmvs_hip.so does not export " + n
    assert sorted(binding.SYMBOLS) == names  # the ctypes binding covers the whole header


def test_library_does_not_link_or_reference_the_oracle():
    out = subprocess.check_output(["readelf", "-d", binding.LIB_PATH]).decode()
    assert "oracle" not in out
    syms = subprocess.check_output(["nm", "-D", binding.LIB_PATH]).decode()
    assert "hcor_" not in syms
    for f in glob.glob(os.path.join(ROOT, "hc-mvs_amd", "**", "*"), recursive=True):
        if f.endswith((".py", ".cpp", ".hip", ".h")):
            txt = open(f).read()
            for needle in ("oracle/", "oracle_lib", "hcor_", "libhcmvs_oracle", "portable_math.h"):
                assert needle not in txt, "%s references the oracle (%s)" % (f, needle)


def test_default_params_are_the_reference_defaults():
    p = binding.default_params()
    # DepthMap.cpp:117-124, DensifyPointCloud.cpp:163
    assert (p.adapthalfwin, p.n_random_iters, p.propagate_halfwin, p.propagate_step) == (5, 6, 1, 4)
    assert abs(p.ncc_threshold_keep - 0.55) < 1e-7 and abs(p.random_depth_ratio - 0.003) < 1e-9
    assert (p.random_angle1_deg, p.random_angle2_deg, p.random_smooth_normal_deg) == (16.0, 10.0, 13.0)


def test_no_gpu_means_no_compute():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(binding.HcmvsError) as e:
        binding.Context(0)
    assert e.value.code == binding.ERR_NO_DEVICE


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """The boundary is a C ABI (what a cgo / JNI / ctypes stub would bind): the header compiles as C99 with nothing but the C
    library's headers, and a C program links against the shared library and calls through it (no compute without a GPU)."""
    src = tmp_path / "c_client.c"
    src.write_text(r'''
#include "hcmvs_hip.h"
#include <stdio.h>
#include <string.h>
int main(void) {
    hcmvs_ctx* ctx = NULL;
    hcmvs_params p;
    hcmvs_default_params(&p);
    if (p.n_random_iters != 6) return 2;
    int rc = hcmvs_create(0, &ctx);
    if (rc == HCMVS_OK) {               /* a GPU is present: one trivial call through the context */
        if (!hcmvs_last_error(ctx)) return 3;
        hcmvs_destroy(ctx);
    } else if (rc != HCMVS_ERR_NO_DEVICE && rc != HCMVS_ERR_HIP) {
        return 4;
    }
    printf("c client ok %d\n", rc);
    return 0;
}
''')
    exe = tmp_path / "c_client"
    libdir = os.path.dirname(binding.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", libdir, "-lhcmvs_hip", "-Wl,-rpath," + libdir])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "c client ok" in out.stdout, out.stdout + out.stderr
