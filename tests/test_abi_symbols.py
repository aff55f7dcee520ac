"""CPU: libholo_mi355x.so builds for gfx950, loads, and exports every entry point include/holo_abi.h declares
(no compute calls are made: there is no GPU here)."""
import ctypes
import os
import re
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(REPO, "holo_diffusion_amd", "libholo_mi355x.so")
HDR = os.path.join(REPO, "include", "holo_abi.h")


def declared_functions():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(holo_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        subprocess.run(["make", "-C", os.path.join(REPO, "holo_diffusion_amd", "csrc"), "-j8"], check=True,
                       capture_output=True)
    return ctypes.CDLL(LIB)


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ("holo_unet_forward", "holo_ddpm_step", "holo_render", "holo_unet_set_param", "holo_last_error"):
        assert must in names


def test_every_declared_symbol_is_exported(lib):
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} declared in holo_abi.h but not exported"


def test_python_binding_covers_the_header(lib):
    from holo_diffusion_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_functions()
    _lib.bind(lib)
    assert lib.holo_abi_version() == _lib.ABI_VERSION == 6


def test_struct_layouts_match_header():
    from holo_diffusion_amd import _lib
    assert ctypes.sizeof(_lib.HoloUnetCfg) == 4 * (5 + 1 + 8 + 1 + 8 + 2)
    assert ctypes.sizeof(_lib.HoloCamera) == 4 * 16
    assert ctypes.sizeof(_lib.HoloRenderCfg) == 4 * (2 + 1 + 1 + 3 + 2 + 2 + 3 + 1 + 2 + 1 + 1)


def test_error_path_without_gpu(lib):
    from holo_diffusion_amd import _lib
    _lib.bind(lib)
    rc = lib.holo_unet_create(None, None, None)
    assert rc < 0 and b"null" in lib.holo_last_error()
