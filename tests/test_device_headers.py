"""Host checks of device headers that are plain C++: gpsig_amd/csrc/fast_exp.hpp (the table-driven float64 exp of the RBF / Matern
envelopes) against the long-double library
exp, on the host: the header's arithmetic is fma / rint / ldexp only, so the device computes the same bits."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fast_exp_within_one_and_a_half_ulp(tmp_path):
    exe = str(tmp_path / "test_fast_exp")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "gpsig_amd", "csrc"),
                           "-o", exe, os.path.join(ROOT, "tests", "emu", "test_fast_exp.cpp")])
    worst_a, worst_t, worst_1024, worst_2048, worst_2l, edges = subprocess.check_output([exe, "2000000"]).split()
    assert float(worst_a) < 1.5 and float(worst_t) < 1.5, (worst_a, worst_t)
    # the degree-3 variants: 2048 entries lose nothing; 1024 entries (what the Kzx tile kernel takes for incremental tensors) drop a
    # term of 5.5e-16 relative at the ends of the reduction interval: observed 5.8 ulp, nine orders below the 1e-6 parity tolerance
    assert float(worst_2048) < 1.5 and float(worst_1024) < 6.5, (worst_1024, worst_2048)
    # the two-level form of the 1024-entry table (two conflict-free tables of 32 entries): one more rounding, the entries' product
    assert float(worst_2l) < 7.5, worst_2l
    assert int(edges) == 1


def test_tile_kernel_level_partition(tmp_path):
    """gpsig_amd/csrc/tvs_plan.hpp: how the Kzx tile kernel deals the signature levels to the waves of a workgroup."""
    exe = str(tmp_path / "test_tvs_plan")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "gpsig_amd", "csrc"), "-o", exe,
                           os.path.join(ROOT, "tests", "emu", "test_tvs_plan.cpp")])
    assert subprocess.check_output([exe]).split() == [b"0"]


def test_contraction_depth_pieces(tmp_path):
    """gpsig_amd/csrc/sig_pieces.hpp: the depth pieces of the feature contraction (equal pieces as in round 3, the last one cut into
    finer ones of halving size in round 4) cover every slab exactly once and depend on the depth and the two counts alone."""
    exe = str(tmp_path / "test_sig_pieces")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "gpsig_amd", "csrc"), "-o", exe,
                           os.path.join(ROOT, "tests", "emu", "test_sig_pieces.cpp")])
    assert subprocess.check_output([exe]).split() == [b"0"]
