"""Invariants of the COMPILED ping-pong GEMM (gfx950 ISA, cross-compiled here: no GPU needed).

Round 3 found two performance bugs that no numerical test can see, both by reading `hipcc -S` output:

* a scratch reload (`scratch_load` + the `s_waitcnt vmcnt(0)` behind it) BETWEEN the DMA issues of a K tile serialises the whole
  prefetch -- the K-extension build of the persistent kernel ran 50 % slower that way;
* spills inside the main loop of a 256-VGPR kernel.

This test compiles csrc/gemm.hip to assembly (cached under csrc/build/, ~100 s when stale) and asserts, for every instantiation of
the persistent kernel `gemm_nt_kernel_v4` that the default step launches, that (a) no scratch instruction sits within ten lines of
a `global_load_lds` and (b) the innermost loop around the MFMAs holds no scratch instruction at all.
"""
import os
import re
import subprocess

import pytest

from tiny_audio_amd import _lib

CSRC = os.path.join(os.path.dirname(os.path.abspath(_lib.__file__)), "csrc")
# template arguments <BN2, ACT, OUT_BF16, HAS_RES, KEXT, LIFE, BM2> of the kernels behind the default step (profiles/r03_final2_kernel_steps.md)
STEP_KERNELS = ["Li320ELi0ELb1ELb1ELb0ELb0ELi256E", "Li320ELi1ELb1ELb0ELb0ELb0ELi256E", "Li320ELi2ELb1ELb0ELb0ELb0ELi256E",
                "Li320ELi0ELb1ELb0ELb0ELb0ELi256E", "Li256ELi0ELb1ELb0ELb0ELb0ELi192E", "Li256ELi0ELb1ELb0ELb0ELb0ELi256E",
                "Li256ELi0ELb0ELb0ELb0ELb0ELi192E"]


def _gemm_isa():
    src = os.path.join(CSRC, "gemm.hip")
    out = os.path.join(CSRC, "build", "gemm_isa.s")
    deps = [src] + [os.path.join(CSRC, h) for h in ("common.h", "internal.h", "gelu_lut.h")]
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(d) for d in deps):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        cmd = [_lib.hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form", "--cuda-device-only",
               "-S", src, "-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
    with open(out) as fh:
        return fh.read()


def _kernels(isa):
    for m in re.finditer(r"^(_Z\d+gemm_nt_kernel_v4I[^\n:]*):.*?; NumVgprs: (\d+).*?; ScratchSize: (\d+)", isa, re.S | re.M):
        yield m.group(1), isa[m.start():m.end()].split("\n"), int(m.group(2)), int(m.group(3))


@pytest.mark.timeout(900)
def test_persistent_gemm_has_no_scratch_traffic_around_its_dma_or_in_its_main_loop():
    isa = _gemm_isa()
    seen = set()
    for name, lines, vgprs, scratch in _kernels(isa):
        key = next((k for k in STEP_KERNELS if "kernel_v4I" + k in name), None)
        if key is None:
            continue
        seen.add(key)
        dma = [i for i, l in enumerate(lines) if "global_load_lds" in l]
        scr = [i for i, l in enumerate(lines) if "scratch_" in l]
        assert dma, name
        near = [lines[i].strip() for i in scr if any(abs(i - d) <= 10 for d in dma)]
        assert not near, (name, near[:3])
        # innermost loop that encloses the MFMAs: the smallest backward branch range around them
        labels = {mm.group(1): i for i, l in enumerate(lines) for mm in [re.match(r"(\.LBB\d+_\d+):", l)] if mm}
        mf = [i for i, l in enumerate(lines) if "v_mfma" in l]
        loops = []
        for i, l in enumerate(lines):
            mm = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
            if mm and mm.group(1) in labels and labels[mm.group(1)] < i and labels[mm.group(1)] <= mf[0] and i >= mf[-1]:
                loops.append((labels[mm.group(1)], i))
        assert loops, name
        a, b = min(loops, key=lambda t: t[1] - t[0])
        inside = [lines[i].strip() for i in scr if a <= i <= b]
        assert not inside, (name, inside[:3])
        assert vgprs <= 256
    assert seen == set(STEP_KERNELS), sorted(set(STEP_KERNELS) - seen)


# ---------------------------------------------------------------------------- round 4: the one-wave-per-SIMD AGPR kernel (gemm_v7.hip)
def _gemm_v7_isa():
    src = os.path.join(CSRC, "gemm_v7.hip")
    out = os.path.join(CSRC, "build", "gemm_v7_isa.s")
    deps = [src] + [os.path.join(CSRC, h) for h in ("common.h", "gemm_common.h", "gelu_lut.h")]
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(d) for d in deps):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        # the product flags of this file (_lib.build): no -amdgpu-mfma-vgpr-form
        cmd = [_lib.hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", src, "-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
    with open(out) as fh:
        return fh.read()


@pytest.mark.timeout(900)
def test_v7_gemm_keeps_its_accumulators_in_agprs_and_its_main_loop_clean():
    """Every instantiation of gemm_nt_kernel_v7<MI, NJ, ...>: no scratch at all, the accumulators in AGPRs (min(256, 4 MI NJ) of
    them), and the k loop -- the innermost loop around the MFMAs -- is exactly MI x NJ MFMAs on AGPR / VGPR accumulators in place,
    MI + NJ fragment reads, (MI + NJ) / 2 DMA pieces, one counted vmcnt wait + one barrier, and not a single accumulator move."""
    isa = _gemm_v7_isa()
    pat = re.compile(r"^(_Z\d+gemm_nt_kernel_v7ILi(\d+)ELi(\d+)E[^\n:]*):.*?; NumVgprs: (\d+)\n; NumAgprs: (\d+)\n; TotalNumVgprs: (\d+)\n"
                     r"; ScratchSize: (\d+)\n.*?; Occupancy: (\d+)", re.S | re.M)
    seen = 0
    for m in pat.finditer(isa):
        name, mi, nj, agprs, scratch, occ = m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(5)), int(m.group(7)), int(m.group(8))
        lines = isa[m.start():m.end()].split("\n")
        seen += 1
        assert scratch == 0, (name, scratch)
        assert occ == 1 and agprs >= min(256, 4 * mi * nj), (name, agprs, occ)
        assert not any("scratch_" in l for l in lines), name
        labels = {mm.group(1): i for i, l in enumerate(lines) for mm in [re.match(r"(\.LBB\d+_\d+):", l)] if mm}
        mf = [i for i, l in enumerate(lines) if l.strip().startswith("v_mfma")]
        assert len(mf) == mi * nj, (name, len(mf))              # one copy of the k-step body
        loops = []
        for i, l in enumerate(lines):
            mm = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
            if mm and mm.group(1) in labels and labels[mm.group(1)] < i and labels[mm.group(1)] <= mf[0] and i >= mf[-1]:
                loops.append((labels[mm.group(1)], i))
        assert loops, name
        a, b = min(loops, key=lambda t: t[1] - t[0])
        body = [l.strip() for l in lines[a:b + 1]]
        assert not any("v_accvgpr" in l for l in body), name
        for l in body:
            if l.startswith("v_mfma"):                            # in place: vdst == src2
                ops_ = [o.strip() for o in l.split(None, 1)[1].split(",")]
                assert ops_[0] == ops_[3], (name, l)
        assert sum(l.startswith("ds_read_b128") for l in body) == mi + nj, name
        assert sum("global_load_lds_dwordx4" in l for l in body) == (mi + nj) // 2, name
        assert sum(l.startswith("s_barrier") for l in body) == 1, name
        waits = [l for l in body if l.startswith("s_waitcnt") and "vmcnt" in l]
        assert waits == [f"s_waitcnt vmcnt({(mi + nj) // 2})"], (name, waits)
    assert seen >= 18, seen
