"""K3's two prefix edit distance searches (dream_yara_amd/csrc/myers_device.h: the multi-word block form and the one-word banded form,
u32 for up to 15 errors and u64 for up to 31) compiled for the host and checked against a plain dynamic program on random, low-complexity,
truncated and N-bearing inputs: the sequence of (end position, distance) reports must be identical."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_banded_and_block_myers_equal_the_plain_dp(tmp_path):
    exe = str(tmp_path / "myers_fuzz")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "native", "myers_fuzz.cpp")])
    out = subprocess.check_output([exe, "150000"]).decode()
    assert out.startswith("ok banded=150000 block=150000"), out


def test_closed_extension_sides_equal_the_searches(tmp_path):
    """K3's first pass closes an extension side that holds one substitution without the edit distance search (extend_sides.h: closeLeftSide /
    closeRightSide); wherever the closed form applies, its end position and its cost of 1 must be what the search (the second pass's code,
    same header) answers — random, low-complexity (the shifted end ties), N-bearing and clipped windows, with an address sanitizer watching
    the 8-byte loads."""
    exe = str(tmp_path / "extend_fuzz")
    subprocess.check_call(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-std=c++17", "-o", exe,
                           os.path.join(ROOT, "tests", "native", "extend_fuzz.cpp")])
    out = subprocess.check_output([exe, "150000"]).decode()
    assert out.startswith("ok: "), out
