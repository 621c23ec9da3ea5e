"""Register budgets of the two kernels the BASELINE configurations run on (no GPU needed: hipcc cross-compiles gfx950).

Round 3 measured what a spilled vector register costs these kernels ("the register diet": the interior
+17 %, the material balls +16 % from 121 -> 12 spilled registers at four wavefronts per SIMD), so the budgets are part of the
contract: the Cornell packet kernel fits 128 VGPRs (four wavefronts per SIMD) without spilling, the phase machine of configs
3 / 4 fits them with at most 20 spilled (12 until round 4's pinned tree pointers, which cost eight more in the shade body and
still measured +1.5 % on the material balls: gpurun r4e). tools/probe_*.hip instantiate exactly those kernels from the product's headers
(~12 s each); tools/kernel_resources.py prints the table for every variant."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-gpu-flush-denormals-to-zero", "-c",
         "-Rpass-analysis=kernel-resource-usage"]


def _resources(probe, kernel, tmp_path, *defs):
    out = subprocess.run([HIPCC] + FLAGS + list(defs) + [os.path.join(ROOT, "tools", probe), "-o", str(tmp_path / "probe.o")],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    for blk in re.split(r"remark: Function Name: ", out.stderr)[1:]:
        if kernel in blk.split()[0]:
            val = lambda key: int(re.search(re.escape(key) + r": (\d+)", blk).group(1))
            return {"vgprs": val("VGPRs"), "spilled": val("VGPRs Spill"), "waves": val("Occupancy [waves/SIMD]")}
    raise AssertionError("kernel %s not in the remarks of %s" % (kernel, probe))


pytestmark = pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="needs hipcc")


def test_cornell_packet_kernel_fits_five_waves_with_no_scratch_in_its_inner_loops(tmp_path):
    """The headline kernel (BASELINE configs[1]): since round 6 compiled for FIVE wavefronts per SIMD — 96 registers, at most 19 values in scratch, stored once in
    front of the pixel loop and reloaded in its body, never inside the leaf-box or candidate loops (resident_kernel.h: MIW_PACKET_WAVES; 248.1 -> 238.3 ms, gpurun r6n).
    At four (-DMIW_PACKET_WAVES=4: rounds 3 - 5) it fits 128 registers without any."""
    r = _resources("probe_resident.hip", "k_path_residentILb1ELi2ELi1ELb0ELj0E", tmp_path)
    assert r["waves"] == 5 and r["vgprs"] <= 96 and r["spilled"] <= 19, r      # (12 before the chunk jobs of QueueWork::fetch_job; 239.4 -> 221.9 ms with them, gpurun r6p - r6s)
    out = subprocess.run([HIPCC] + [f for f in FLAGS if not f.startswith("-Rpass")] + ["-S", os.path.join(ROOT, "tools", "probe_resident.hip"),
                          "--cuda-device-only", "-o", str(tmp_path / "probe.s")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = open(tmp_path / "probe.s").read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z15k_path_residentILb1ELi2ELi1ELb0ELj0E"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    depth = ""
    inner = total = 0
    for l in lines[start:end]:
        m = re.match(r"^\.LBB\d+_\d+:\s*;?(.*)", l)
        if m:
            depth = m.group(1)
        if "scratch_" in l:
            total += 1
            inner += 0 if re.search(r"Depth=1\b", depth) or "Depth=" not in depth else 1
    assert total <= 32 and inner == 0, (total, inner)
    r = _resources("probe_resident.hip", "k_path_residentILb1ELi2ELi1ELb0ELj0E", tmp_path, "-DMIW_PACKET_WAVES=4")
    assert r["waves"] == 4 and r["vgprs"] <= 128 and r["spilled"] == 0, r


def test_phase_machine_of_configs_3_and_4_fits_four_waves(tmp_path):
    r = _resources("probe_phased.hip", "k_path_phasedILi3ELb0ELb1ELi4ELi2E", tmp_path, "-DMIW_PROBE_C34=1")     # over the 8-wide tree: what runs since round 5
    # (8 spilled without the second pending triangle group, 24 with it — MIW_W8_SPEC, the default: measured +2.7 % / +3.5 % on configs 3 / 4
    # over the non-speculating walk WITH these spills, gpurun r5b / r5d)
    # (round 6: the chunk jobs of QueueWork::fetch_job compiled in — 48 / 24; without them, -DMIW_PHASED_JOBS=0, the 24 / 16 of round 5)
    assert r["waves"] == 4 and r["vgprs"] <= 128 and r["spilled"] <= 48, r
    r = _resources("probe_phased.hip", "k_path_phasedILi3ELb0ELb1ELi4ELi1E", tmp_path, "-DMIW_PROBE_C34=1")     # its 4-wide twin (MIW_BVH8=0, trees the 8-wide collapse refuses)
    assert r["waves"] == 4 and r["vgprs"] <= 128 and r["spilled"] <= 24, r
    r = _resources("probe_phased.hip", "k_path_phasedILi3ELb0ELb1ELi4ELi2E", tmp_path, "-DMIW_PROBE_C34=1", "-DMIW_PHASED_JOBS=0")
    assert r["waves"] == 4 and r["vgprs"] <= 128 and r["spilled"] <= 24, r


def test_film_replay_keeps_its_sample_loops_free_of_scratch(tmp_path):
    """k_film_lanes (device/film_kernels.h): 80 sums + two buffers of record words per lane, compiled for three wavefronts per SIMD.
    What does not fit 168 registers is moved around the 30 specialised sample loops (once per pixel step of ~500 samples), never inside
    one: every innermost loop of the kernel's ISA is free of scratch_ instructions."""
    out = subprocess.run([HIPCC] + [f for f in FLAGS if not f.startswith("-Rpass")] + ["-S", os.path.join(ROOT, "tools", "probe_film.hip"),
                          "--cuda-device-only", "-o", str(tmp_path / "probe.s")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = open(tmp_path / "probe.s").read().split("\n")
    # ... and (round 6) the form that runs BESIDE the path kernel — two records in flight per lane, compiled for four wavefronts per SIMD so that a
    # wavefront of it fits next to three of the 120-register packet kernel (miwave.hip: overlap_enqueue): 128 registers, the same property
    for name, cap, records in (("_Z12k_film_lanesILi4ELi0ELi3E", 168, 4), ("_Z12k_film_lanesILi2ELi1ELi4E", 128, 2)):
        _film_lanes_loops_are_free_of_scratch(lines, name, cap, records)


def _film_lanes_loops_are_free_of_scratch(lines, name, cap, records):
    start = next(i for i, l in enumerate(lines) if l.startswith(name))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end]
    meta = "\n".join(lines)
    blk = meta[meta.index(".name:           " + name):]
    assert int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1)) <= cap
    loops = 0
    for i, l in enumerate(body):
        if "Inner Loop Header: Depth=2" not in l:
            continue
        j = i
        while not re.match(r"^\.LBB\d+_\d+:", body[j]):
            j -= 1
        label = body[j].split(":")[0]
        k = j + 1
        while not ("s_cbranch" in body[k] and label in body[k]):
            k += 1
        assert not any("scratch_" in b for b in body[j:k]), "scratch traffic inside the sample loop at %s" % label
        assert sum("v_pk_add_f32" in b for b in body[j:k]) >= 5 * records        # (it is a sample loop: a trip's samples x one row x one column pair x five channels)
        loops += 1
    assert loops == 30                                                           # 10 row ranges x 3 column-pair ranges
