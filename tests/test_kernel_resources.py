"""Register budgets of the hot kernels at d = 128 (CH = 2), read from hipcc's own resource report
(cross-compiled: no GPU needed).  Round 3 found cbow_cached_kernel sitting ON the 128-VGPR cliff:
a refactoring that cost 4 registers took it from 4 to 3 waves per SIMD and from 0.86 to 0.66 of
the roofline -- nothing else had changed.  These kernels are latency bound; occupancy is their
throughput.  The numbers below are the budgets the measured profiles were taken with."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "embiggen_amd", "csrc")


_CACHE = {}


def resources(unit, tmp_path):
    if unit not in _CACHE:
        _CACHE[unit] = _resources(unit, tmp_path)
    return _CACHE[unit]


def _resources(unit, tmp_path):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    res = subprocess.run(
        [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-c",
         "-Rpass-analysis=kernel-resource-usage", os.path.join(CSRC, unit), "-o",
         str(tmp_path / "unit.o")], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    out, cur = {}, None
    for line in res.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
        for key, short in (("VGPRs", "vgprs"), ("ScratchSize [bytes/lane]", "scratch"),
                           ("Occupancy [waves/SIMD]", "waves")):
            m = re.search(re.escape(key) + r": (\d+)", line)
            if m and cur is not None:
                cur[short] = int(m.group(1))
    return out


def pick(table, fragment):
    hits = {k: v for k, v in table.items() if fragment in k}
    assert hits, fragment
    return hits


@pytest.mark.timeout(1200)
def test_walk_ordered_kernels_keep_their_occupancy(tmp_path):
    table = resources("gn2v_api.hip", tmp_path)
    # cbow_cached_kernel<CH = 2, write-through / write-back>: 4 waves per SIMD, no scratch
    for name, r in pick(table, "cbow_cached_kernelILi2E").items():
        assert r["vgprs"] <= 128 and r["waves"] >= 4 and r["scratch"] == 0, (name, r)
    # cbow_lazy_kernel (the CBOW default): capped at 4 waves per SIMD, at most a handful of spills
    for name, r in pick(table, "cbow_lazy_kernelILi2E").items():
        assert r["vgprs"] <= 128 and r["waves"] >= 4 and r["scratch"] <= 32, (name, r)
    # rows of 132-256 floats (CH = 4): three workgroups per CU, two for rows of exactly 256 (FULL;
    # its window takes 56 KB of LDS): no 128-register cap there, next to no spills
    for name, r in pick(table, "cbow_lazy_kernelILi4E").items():
        full = name.endswith("Lb1EEEvNS_9TrainArgsE")
        assert r["waves"] >= (2 if full else 3) and r["scratch"] <= 32, (name, r)
    for name, r in pick(table, "sgns_cached_kernelILi2E").items():
        assert r["vgprs"] <= 102 and r["waves"] >= 5 and r["scratch"] == 0, (name, r)
    # d = 256: the cached SkipGram kernel is capped at 4 waves (2 registers spilled), GloVe runs at 4
    for name, r in pick(table, "sgns_cached_kernelILi4E").items():
        assert r["vgprs"] <= 128 and r["waves"] >= 4 and r["scratch"] <= 16, (name, r)
    for wm in ("Li0E", "Li1E"):  # the store flavours (the atomic one: 2 waves, as ever)
        for name, r in pick(table, "glove_kernelILi8E" + wm).items():
            assert r["waves"] >= 4 and r["scratch"] == 0, (name, r)
    # otherwise the training kernels never spill at CH <= 8 (d <= 512)
    for name, r in table.items():
        if re.search(r"(sgns|cbow)(_cached)?_kernelILi[1248]E", name) and "sgns_cached_kernelILi4E" not in name:
            assert r["scratch"] == 0, (name, r)


@pytest.mark.timeout(1200)
def test_block_kernel_keeps_its_occupancy(tmp_path):
    table = resources("gn2v_block_api.hip", tmp_path)
    # sgns_block_kernel<CH = 2, contextual write-back, central atomic, parallel>: the bench's kernel
    # (its 80 spilled SGPRs overflow the VGPR lanes into 24 B of scratch per lane: the two paths of
    # train_record keep many uniform values alive; measured harmless, bounded here)
    for name, r in pick(table, "sgns_block_kernelILi2ELi1ELi2ELb0E").items():
        assert r["vgprs"] <= 128 and r["waves"] >= 4 and r["scratch"] <= 32, (name, r)
    for name, r in table.items():
        if re.search(r"sgns_block_kernelILi[1248]E", name):
            assert r["scratch"] <= 64, (name, r)


@pytest.mark.timeout(1200)
def test_resident_kernel_keeps_its_occupancy(tmp_path):
    """sgns_resident_v2_kernel<CH = 2>: sixteen waves per workgroup, one workgroup per CU -- four
    waves per SIMD means at most 128 registers, and the bench's kernel spills none."""
    table = resources("gn2v_block_api.hip", tmp_path)
    seen = 0
    for name, r in pick(table, "sgns_resident_v2_kernelILi2E").items():
        if "Lb0ELi16EEE" in name:  # the parallel form (not the in-order one), sixteen waves
            assert r["vgprs"] <= 128 and r["waves"] >= 4 and r["scratch"] == 0, (name, r)
            seen += 1
    assert seen == 3  # strides of 96 and 128 floats as constants, and the run-time stride
    # rows of 129-512 floats: eight waves per workgroup, 256 registers a lane; no spills where the
    # stride is a compile-time constant (multiples of 64 floats)
    for ch, ldqs in ((4, (6, 8)), (8, (10, 12, 14, 16))):
        for ldq in ldqs:
            one = pick(table, f"sgns_resident_v2_kernelILi{ch}ELi{ldq}ELb0ELi8EEE")
            assert len(one) == 1
            for name, r in one.items():
                assert r["waves"] >= 2 and r["scratch"] == 0, (name, r)


@pytest.mark.timeout(1200)
def test_walk_kernels_keep_their_occupancy(tmp_path):
    """The record sampler runs three waves per SIMD (measured: as fast as four, which costs
    spills); its scratch is the frame of the rare exact scan, not spills of the trial loop."""
    table = resources("gn2v_api.hip", tmp_path)
    for name, r in pick(table, "walk_rec_kernel").items():
        # (typed AND sub-sampled -- type factors on a graph with rows longer than max_neighbours --
        # spills a dozen registers of the trial loop: 392 B; the untyped kernels of the bench: 304)
        limit = 400 if "ILb1ELb1E" in name else 352
        assert r["waves"] >= 3 and r["scratch"] <= limit, (name, r)
    for name, r in pick(table, "walk_rec_kernelILb0E").items():
        assert r["scratch"] <= 304, (name, r)  # the frame of the out-of-line exact scan, no spills
    for name, r in pick(table, "walk_kernelILb").items():
        assert r["waves"] >= 3 and r["scratch"] <= 320, (name, r)
