"""Register / scratch / occupancy table of every gfx950 kernel in libmiwave (no GPU needed).

    python tools/kernel_resources.py [scalar_rgb|scalar_spectral] [filter-substring]

Compiles csrc/miwave.hip with -Rpass-analysis=kernel-resource-usage (same flags as mitsuba2_amd/build.py)
and prints one line per kernel: a quick check after touching a leaf header that the hot kernels still fit
their launch bounds without spilling."""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mitsuba2_amd import build  # noqa: E402


def main():
    variant = sys.argv[1] if len(sys.argv) > 1 else "scalar_rgb"
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    src = os.path.join(build.PKG, "csrc", "miwave.hip")
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [build.HIPCC] + build.HIP_FLAGS + build.VARIANTS[variant][1] + \
              [src, "-o", os.path.join(tmp, "x.so"), "-Rpass-analysis=kernel-resource-usage"]
        text = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows = []
    for blk in re.split(r"remark: Function Name: ", text)[1:]:
        name = blk.split()[0]
        val = lambda key: int(re.search(re.escape(key) + r": (\d+)", blk).group(1))
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = dem.split("(")[0].replace("void ", "")
        if flt in dem:
            rows.append((dem, val("VGPRs"), val("AGPRs"), val("VGPRs Spill"), val("SGPRs Spill"), val("ScratchSize [bytes/lane]"),
                         val("Occupancy [waves/SIMD]"), val("LDS Size [bytes/block]")))
    print("%-64s %5s %5s %7s %7s %8s %4s %7s" % ("kernel", "VGPR", "AGPR", "v-spill", "s-spill", "scratch", "occ", "LDS"))
    for r in rows:
        print("%-64s %5d %5d %7d %7d %8d %4d %7d" % r)


if __name__ == "__main__":
    main()
