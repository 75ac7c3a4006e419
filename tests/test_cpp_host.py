"""The C++ host-side mirror (include/mcptam_hip/ChainBundle.hpp) compiles with plain g++ against the C ABI and links
against libmcptam_hip.so; on a GPU box the same program solves a small bundle."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp, name="chain_bundle_smoke"):
    import __graft_entry__ as g
    g.build()
    exe = os.path.join(str(tmp), name)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", name + ".cpp"),
                           "-L", os.path.join(ROOT, "mcptam_amd"), "-lmcptam_hip", "-Wl,-rpath," + os.path.join(ROOT, "mcptam_amd"),
                           "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


def test_cpp_mirror_compiles_and_links(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([exe, "--link-only"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "linked" in out.stdout


@pytest.mark.gpu
def test_cpp_mirror_solves_on_gpu(tmp_path, gpu_required):
    exe = _build(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr


def test_cpp_keyframe_mirror_compiles_and_links(tmp_path):
    """include/mcptam_hip/KeyFrame.hpp (KeyFrame / Level, SmallBlurryImage, Relocaliser scoring, MiniPatch and the tracker's
    per-point entry points): every member is instantiated by tests/cpp/keyframe_link.cpp and must link against the C ABI."""
    exe = _build(tmp_path, "keyframe_link")
    out = subprocess.run([exe, "--link-only"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "linked" in out.stdout


@pytest.mark.gpu
def test_cpp_keyframe_mirror_runs_on_gpu(tmp_path, gpu_required):
    """The same program on a GPU box: pyramids / corners / LUTs of two KeyFrames fed the same frame agree, self-alignment of the
    small blurry images is the identity, the relocaliser score is 0, MiniPatch matches sit at zero offset, the frame history
    advances, and a pose update from perfect measurements is zero (tests/cpp/keyframe_link.cpp)."""
    exe = _build(tmp_path, "keyframe_link")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("keyframe mirror ok") == 2 and "one submission" in out.stdout       # 320x240 and 640x480, batch entries included
    print(out.stdout)
