"""CPU tier: the hand-counted `s_waitcnt vmcnt(n)` of the tiled kernel against the ISA hipcc really emitted (cross-compiled
for gfx950, no GPU needed).  See tools/isa/check_vmcnt.py for what is counted and why a too-large count is harmless and a
too-small one is a silent race."""
import os
import subprocess

from conftest import ROOT


def test_counted_vmcnt_wait_matches_the_emitted_stores():
    csrc = os.path.join(ROOT, "rust-seq2kminmers_amd", "csrc")
    r = subprocess.run(["make", "-C", csrc, "-j", "2", "isa-check"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "vector-memory ops per round" in r.stdout


def test_tiled_kernels_fit_their_waves_per_simd_without_scratch():
    """The persistent tile kernel runs one block per CU: 12 waves = three per SIMD for Hpc -- 512 VGPRs / 3 = 168 per wave by its own occupancy, but the
    k-min-mer kernel's waves (57 VGPRs) run BESIDE it and need the rest: at 159 the step was 11 % slower (profiles/r05_ab_args_in_vgprs.txt), so its budget is
    128 --, 16 = four per SIMD for the Regular family (128), and no
    scratch (a spill in the hash loop costs more than anything else in it).  Round 3 lost 20-55 VGPRs to loop-invariant lane
    arithmetic hoisted across the hash loop; this keeps that from coming back unnoticed.  Also: the compile-time-l
    kernels use gfx950's three-input bit operation in the hash loop (2 x 144 positions x 2 strands and more)."""
    import re

    csrc = os.path.join(ROOT, "rust-seq2kminmers_amd", "csrc")
    r = subprocess.run(["make", "-C", csrc, "-j", "2", "isa/s2k_tile_L31.s", "isa/s2k_tile_L0.s"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    seen = 0
    for name in ("s2k_tile_L31.s", "s2k_tile_L0.s"):
        text = open(os.path.join(csrc, "isa", name)).read()
        for m in re.finditer(r"\.name:\s+(\S*tile_minimizer_kernel\S*)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", text):
            kern, vgprs, spills = m.group(1), int(m.group(2)), int(m.group(3))
            seen += 1
            # Hpc kernels: three waves per SIMD and two k-min-mer waves beside them; Regular family: four waves per SIMD (16-wave blocks): <= 128 either way
            assert vgprs <= 128, (name, kern, vgprs)
            assert spills == 0, (name, kern, spills)
        for m in re.finditer(r"; ScratchSize: (\d+)", text):
            assert int(m.group(1)) == 0, (name, "scratch", m.group(1))
        if name == "s2k_tile_L31.s":
            assert text.count("bitop3:0x96") >= 4 * 2 * 144, "v_bitop3_b32 (three-input XOR) missing from the hash loop"
    assert seen >= 8, seen  # 4 instantiations (Hpc / Regular x descriptor / legacy) per translation unit


def test_profile_and_knobs_configurations_still_compile():
    """`make PROFILE=1` / `make KNOBS=1` are what tools/phases.sh and tools/ab/*.sh build on the GPU box; a static_assert that only
    fires in those configurations (round 5: the PROFILE accumulators pushed the Regular block over 160 KiB of LDS) would go unnoticed
    until a GPU call is spent on it.  Syntax-only, host + device passes, a few seconds each."""
    csrc = os.path.join(ROOT, "rust-seq2kminmers_amd", "csrc")
    # (-DS2K_STREAM_BUILD=1: the Regular kernel without a tile buffer, measured slower and left out of the default build -- profiles/r06_stream.txt)
    for flags, src in ((["-DS2K_PROFILE", "-DS2K_DEBUG_KNOBS"], "s2k_tile.hip"), (["-DS2K_DEBUG_KNOBS"], "s2k_tile.hip"),
                       (["-DS2K_STREAM_BUILD=1", "-DS2K_TILE_L=31"], "s2k_tile_inst.hip"),
                       (["-DS2K_LPH=1", "-DS2K_TILE_L=31"], "s2k_tile_inst.hip")):  # (one lane per hit in the listing: measured slower, off by default)
        r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fsyntax-only", *flags, src],
                           cwd=csrc, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (flags, r.stderr[-3000:])


def test_experiment_benches_still_compile():
    """tools/experiments/hash_loop_bench.hip and hash_stream_bench.hip include the kernel's own header (profiles/README cites them as
    reproducible evidence): they must keep compiling against it."""
    for src in ("hash_loop_bench.hip", "hash_stream_bench.hip"):
        r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fsyntax-only", "-I", "rust-seq2kminmers_amd/csrc",
                            os.path.join("tools", "experiments", src)], cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (src, r.stderr[-3000:])
