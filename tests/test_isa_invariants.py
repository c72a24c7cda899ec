"""Invariants of the COMPILED ping-pong GEMM (gfx950 ISA, cross-compiled here: no GPU needed).

Round 3 found two performance bugs that no numerical test can see, both by reading `hipcc -S` output:

* a scratch reload (`scratch_load` + the `s_waitcnt vmcnt(0)` behind it) BETWEEN the DMA issues of a K tile serialises the whole
  prefetch -- the K-extension build of the persistent kernel ran 50 % slower that way;
* spills inside the main loop of a 256-VGPR kernel.

This test compiles csrc/gemm.hip to assembly (cached under csrc/build/, ~100 s when stale) and asserts, for every instantiation of
the persistent kernel `gemm_nt_kernel_v4` that the default step launches, that (a) no scratch instruction sits within ten lines of
a `global_load_lds`, (b) the innermost loop around the MFMAs holds no scratch instruction at all and (c), since round 5, that the
kernel has no scratch at all (`ScratchSize == 0`); the one-tile-per-CU kernels of the step (`gemm_nt_kernel_v5`) likewise.
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
        # round 5: NO scratch at all in the kernels of the default step.  The last one was the 256-VGPR
        # o_proj / fc2 kernel <320, 0, true, true>: 8 bytes -- the reciprocal of the C row map's divisor, hoisted out of the persistent
        # tile loop and reloaded at the top of every epilogue (gemm_common.h: opaque_sgpr)
        assert scratch == 0 and not scr, (name, scratch)
    assert seen == set(STEP_KERNELS), sorted(set(STEP_KERNELS) - seen)


@pytest.mark.timeout(900)
def test_one_tile_per_cu_gemm_of_the_step_has_no_scratch():
    """gemm_nt_kernel_v5<ACT = 0, bf16 out, +/- residual> (the LM's N = 1024 products; profiles/r04_final4_kernel_steps.md)."""
    isa = _gemm_isa()
    seen = 0
    for m in re.finditer(r"^(_Z\d+gemm_nt_kernel_v5ILi0ELb1ELb[01]ELi0ELb0E[^\n:]*):.*?; ScratchSize: (\d+)", isa, re.S | re.M):
        seen += 1
        assert int(m.group(2)) == 0, m.group(1)
    assert seen == 2, seen
