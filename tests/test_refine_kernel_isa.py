"""CPU (hipcc cross-compiles): what the refinement kernel of the RANSAC recording stage may and may not contain, checked on
its ISA.  Round 4's streaming kernel synchronised its waves through LDS spin locks (compare-and-swap / exchange loops with
s_sleep) and stalled about twice in 10^4 small launches beside context churn; round 5's kernel lets a wave
wait for another wave in exactly one way -- s_barrier.  No sleep, no compare-and-swap, no exchange, no polling loop; the only
returning LDS atomic is the scoring ticket (ds_add_rtn); no scratch."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    if not os.path.exists(HIPCC) and shutil.which("hipcc") is None:
        pytest.skip("no hipcc")
    out = tmp_path_factory.mktemp("isa") / "ransac_split.s"
    csrc = os.path.join(ROOT, "rgbdslam_v2_amd", "csrc")
    subprocess.run([HIPCC if os.path.exists(HIPCC) else "hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                    "-fno-fast-math", "-fno-slp-vectorize", "-I", os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
                    "-o", str(out), os.path.join(csrc, "ransac_split.hip")], check=True, capture_output=True, timeout=900)
    text = open(out).read()
    m = re.search(r"^_ZN6rgbdfe20ransac_refine_kernel\w*:(.*?)^\s*s_endpgm.*?\.amdhsa_kernel", text, re.S | re.M)
    assert m, "ransac_refine_kernel not found in the ISA"
    return text, text[m.start():text.index(".amdhsa_kernel", m.start())]


def test_waves_wait_for_each_other_at_barriers_only(isa):
    _, body = isa
    ops = re.findall(r"^\s+([a-z_0-9]+)", body, re.M)
    assert "s_barrier" in ops
    for forbidden in ("s_sleep", "ds_cmpst_rtn_b32", "ds_cmpst_b32", "ds_wrxchg_rtn_b32", "ds_cmpst_rtn_b64", "s_sethalt"):
        assert forbidden not in ops, forbidden
    # global memory: the unit counter's fetch-add is the only atomic; LDS: the scoring ticket is the only returning one
    glob_atomics = sorted({o for o in ops if o.startswith("global_atomic") or o.startswith("flat_atomic")})
    assert glob_atomics == ["global_atomic_add"], glob_atomics
    lds_rtn = sorted({o for o in ops if o.startswith("ds_") and "_rtn" in o})
    assert lds_rtn == ["ds_add_rtn_u32"], lds_rtn
    # no loop polls memory for another wave's flag: every LDS / global load that sits in a loop with a sleep is gone with the
    # sleeps; what remains to check is that nothing re-reads ONE address until it changes -- the source has no such construct
    src = open(os.path.join(ROOT, "rgbdslam_v2_amd", "csrc", "ransac_split.hip")).read()
    for gone in ("kSpinBound", "WD_TICK", "gave_up", "qlock", "flag_load", "s_sleep", "atomicCAS", "atomicExch"):
        assert gone not in src, gone


def test_kernel_resources(isa):
    text, _ = isa
    meta = text[text.index("amdhsa.kernels"):]
    k = meta[meta.index("ransac_refine_kernel"):] if "ransac_refine_kernel" in meta else meta
    blocks = re.split(r"\n  - \.", meta)
    refine = [b for b in blocks if "ransac_refine_kernel" in b][0]
    vgpr = int(re.search(r"\.vgpr_count:\s+(\d+)", refine).group(1))
    spill = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", refine).group(1))
    scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", refine).group(1))
    assert vgpr <= 128 and spill == 0 and scratch == 0, (vgpr, spill, scratch)   # 4 waves per SIMD, nothing in scratch
