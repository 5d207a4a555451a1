"""The device quadtree (orb_quadtree in orb.hip) rests on a list-free re-expression of ExtractorNode list surgery.  This test
builds the C++ emulation of that formulation and checks it against the sequential host restatement on 3000 random point sets."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_list_free_formulation_equals_sequential(tmp_path):
    exe = tmp_path / "qt_emul"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", ROOT, os.path.join(ROOT, "tests", "cpp", "quadtree_emul.cpp"), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode()
    assert "all equal" in out, out
