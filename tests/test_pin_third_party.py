"""CPU: tools/pin_third_party/ -- the harness that pins the oracle's restatements of OpenCV / PCL / Eigen arithmetic on the real
libraries.  The libraries do not exist on the build box, so what can be checked here is everything around
them: the harness source compiles (syntax only) against declaration-only stand-ins of the headers it uses; the array container
round-trips between its Python and its C++ implementation; the comparator calls oracle-made pins identical and finds a single
flipped bit."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PIN = os.path.join(ROOT, "tools", "pin_third_party")
sys.path.insert(0, PIN)
import compare_pins  # noqa: E402
import export_inputs  # noqa: E402
import oracle_side  # noqa: E402
import pinfile  # noqa: E402

needs_gxx = pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")


@needs_gxx
def test_harness_source_compiles_against_the_stand_in_headers():
    r = subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-Wall", "-I", os.path.join(PIN, "stubs"),
                        os.path.join(PIN, "pin_third_party.cpp")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    cm = open(os.path.join(PIN, "CMakeLists.txt")).read()
    for need in ("find_package(OpenCV 3", "find_package(Eigen3", "find_package(PCL 1.7", "pin_third_party.cpp"):
        assert need in cm


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    d = tmp_path_factory.mktemp("pins")
    inp = export_inputs.build()
    pinfile.write(str(d / "inputs.pin"), inp)
    pins = oracle_side.oracle_pins(pinfile.read(str(d / "inputs.pin")))
    pinfile.write(str(d / "pins.pin"), pins)
    return d, inp, pins


def test_inputs_cover_every_third_party_call_site(files):
    _, inp, pins = files
    assert int(inp["n_images"][0]) == 3 and inp["img0_gray"].shape == (480, 640) and inp["img2_mask"].min() == 0
    assert list(inp["img0_fast_thresholds"]) == [20, 14, 9]
    # detect at three thresholds per image, retainBest + compute, compute at given keypoints, retainBest with ties, the fits, SVD, LLT
    for name in ("img0_detect0_f", "img2_detect2_octave", "img0_retained_f", "img0_desc", "img0_given_desc", "retain0_kept_f",
                 "fit0_T", "fit16_T", "svd_U", "llt_q"):
        assert name in pins, name
    assert len(pins["img0_detect2_f"]) > len(pins["img0_detect0_f"]) > 600            # a lower threshold finds more
    assert pins["img0_retain_best_size"][0] >= 600 and len(pins["img0_retained_f"]) == 600
    assert pins["img0_desc"].shape[1] == 32 and len(pins["img0_desc"]) == len(pins["img0_described_f"]) <= 600
    assert len(pins["retain0_kept_f"]) > 100                                           # ties at the cut are kept
    assert np.isfinite(pins["fit0_T"]).all() and pins["fit0_T"][3].tolist() == [0.0, 0.0, 0.0, 1.0]
    assert (pins["llt_q"][:48] > 0).all() and pins["llt_q"][-1] == np.finfo(np.float64).max   # the singular matrix: D5


def test_comparator_accepts_oracle_made_pins_and_finds_one_flipped_bit(files):
    d, _, pins = files
    rep = compare_pins.run(str(d / "inputs.pin"), str(d / "pins.pin"))
    assert [r for r in rep if r["status"] not in ("identical", "info")] == []
    assert sum(r["status"] == "identical" for r in rep) == len(pins)
    bad = dict(pins)
    desc = pins["img1_desc"].copy()
    desc[7, 3] ^= 0x10
    bad["img1_desc"] = desc
    T = pins["fit2_T"].copy()
    T.view(np.uint32)[0, 3] ^= 1                                                       # one ulp of a translation
    bad["fit2_T"] = T
    del bad["svd_S"]
    pinfile.write(str(d / "pins_bad.pin"), bad)
    rep = compare_pins.run(str(d / "inputs.pin"), str(d / "pins_bad.pin"))
    wrong = {r["name"]: r for r in rep if r["status"] not in ("identical", "info")}
    assert set(wrong) == {"img1_desc", "fit2_T", "svd_S"}
    assert wrong["img1_desc"]["differing"] == 1 and wrong["img1_desc"]["first_index"] == 7 * 32 + 3
    assert wrong["fit2_T"]["differing"] == 1 and 0 < wrong["fit2_T"]["max_abs_diff"] < 1e-6
    assert wrong["svd_S"]["status"] == "missing in pins"
    r = subprocess.run([sys.executable, os.path.join(PIN, "compare_pins.py"), str(d / "inputs.pin"), str(d / "pins_bad.pin")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and "3 not identical" in r.stdout


@needs_gxx
def test_container_round_trips_through_the_cpp_reader_and_writer(files, tmp_path):
    d, inp, _ = files
    src = tmp_path / "rt.cpp"
    src.write_text('#include "pinfile.hpp"\nint main(int c, char** v) { auto m = pin::read(v[1]); pin::Writer w(v[2]);\n'
                   '  for (auto& kv : m) w.put(kv.first, kv.second.code, kv.second.dims, kv.second.bytes.data()); return 0; }\n')
    exe = tmp_path / "rt"
    subprocess.run(["g++", "-std=c++11", "-O1", "-I", PIN, str(src), "-o", str(exe)], check=True, timeout=120)
    subprocess.run([str(exe), str(d / "inputs.pin"), str(tmp_path / "copy.pin")], check=True, timeout=60)
    back = pinfile.read(str(tmp_path / "copy.pin"))
    assert set(back) == set(inp)
    for k in inp:
        a = np.ascontiguousarray(inp[k])
        assert back[k].dtype == a.dtype and back[k].shape == a.shape and back[k].tobytes() == a.tobytes(), k
