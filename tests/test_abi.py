"""The C-ABI library loads and exports every symbol include/ktlvo.h declares (CPU only: no
compute call is made here, so no GPU is needed)."""
import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "ktlvo.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ktlvo_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(klib):
    names = _declared()
    assert len(names) >= 28
    lib = C.CDLL(klib.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(klib.SIGNATURES) == names     # the binding covers exactly the header


def test_no_torch_types_and_no_oracle_in_product():
    hdr = open(os.path.join(ROOT, "include", "ktlvo.h")).read()
    assert "torch" not in hdr and "at::" not in hdr
    for dirpath, _, files in os.walk(os.path.join(ROOT, "slam_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f)).read()
                assert "ktlvo_oracle" not in txt and "import oracle" not in txt and "orc_" not in txt, f


def test_default_config_matches_reference_constants(klib):
    c = klib.default_config()
    assert (c.width, c.height) == (1242, 375)                       # cameraParameters.h:16-17
    assert (c.fx, c.cx, c.cy, c.bf) == (707.0912, 601.8873, 183.1104, 386.1448)
    assert (c.x_subregions, c.y_subregions, c.max_per_subregion) == (20, 8, 100)
    assert (c.win, c.max_level, c.x_boundary, c.y_boundary) == (21, 4, 30, 30)
    assert (c.ransac_iters, c.anms_keep, c.keyframe_min_tracked) == (300, 1000, 300)
    assert c.ransac_reproj == np.float32(0.1) and c.ransac_conf == float(np.float32(0.99))
    assert klib.lib.ktlvo_strerror(-2).decode().startswith("parameter value")


def test_struct_layouts(klib):
    assert klib.IMAGE_POINT.itemsize == 12 and klib.DEPTH_POINT.itemsize == 24   # pointDefinition.h:18-42
    assert C.sizeof(klib.LkParams) == 40 and C.sizeof(klib.Roi) == 16


def test_create_fails_loudly_on_bad_config_or_missing_device(klib):
    cfg = klib.default_config(width=0)
    assert not klib.lib.ktlvo_create(C.byref(cfg), 0)
    import torch
    if not torch.cuda.is_available():
        try:
            klib.Context()
        except RuntimeError:
            pass
        else:
            raise AssertionError("Context() must raise without a GPU: there is no CPU fallback")


def test_host_mirror_header_is_plain_cxx11_and_keeps_the_reference_names(tmp_path):
    import subprocess
    hdr = os.path.join(ROOT, "include", "ktlvo_host.hpp")
    tu = tmp_path / "tu.cpp"
    tu.write_text('#include "ktlvo_host.hpp"\nint main() { return sizeof(ktlvo_host::Frontend) > 0 ? 0 : 1; }\n')
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I",
                           os.path.join(ROOT, "include"), str(tu)])
    txt = open(hdr).read()
    for name in ("calcOpticalFlowPyrLK", "getOpticalFlow", "adaptiveNonMaximalSuppresion", "stereoImageCallback",
                 "UnprojectStereo", "estimatePoseRANSAC", "pointCloudCallback"):
        assert re.search(r"\b%s\s*\(" % name, txt), name
    assert not re.search(r"#include\s*<(hip|opencv)", txt)


def test_cxx_node_fails_loudly_without_a_device(tmp_path):
    import subprocess
    import torch
    node = os.path.join(ROOT, "tests", "cpp", "frontend_node")
    if not os.path.exists(node):
        subprocess.check_call(["make", "-C", os.path.dirname(node)])
    frames = tmp_path / "z.bin"
    frames.write_bytes(bytes(64 * 48 * 2 * 2))
    r = subprocess.run([node, "only-one-argument"], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr
    if not torch.cuda.is_available():
        r = subprocess.run([node, str(frames), "64", "48", "2", str(tmp_path / "o.bin")], capture_output=True, text=True)
        assert r.returncode == 3 and "ktlvo_create failed" in r.stderr, r.stderr
