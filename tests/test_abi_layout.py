"""The drop-in boundary is a C ABI (include/miwave.h); the Python side mirrors its structs with ctypes (mitsuba2_amd/_capi.py). A
mirror that drifts — a field appended to the header and not to the ctypes class — reads garbage without any error, so the layouts
are compared here: a C program that includes the header prints sizeof and every field's offsetof, gcc compiles it (no GPU, no
library needed), and each ctypes Structure must agree field by field."""
import ctypes as C
import os
import subprocess

from conftest import ROOT


def test_ctypes_mirrors_match_the_header(tmp_path):
    from mitsuba2_amd import _capi
    names = ["mi_texture", "mi_bsdf", "mi_shape", "mi_emitter", "mi_envmap", "mi_rectangle", "mi_sphere", "mi_bitmap", "mi_scene_desc",
             "mi_rays_soa", "mi_hits_soa", "mi_surface_interaction", "mi_direction_sample", "mi_render_cfg", "mi_counters"]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "miwave.h"', 'int main(void) {']
    for n in names:
        cls = getattr(_capi, n)
        lines.append('  printf("%s sizeof %%zu\\n", sizeof(%s));' % (n, n))
        for f in cls._fields_:
            lines.append('  printf("%s %s %%zu\\n", offsetof(%s, %s));' % (n, f[0], n, f[0]))
    lines += ['  return 0;', '}']
    src = tmp_path / "abi.c"; src.write_text("\n".join(lines))
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    seen = 0
    for line in out.splitlines():
        n, field, value = line.split()
        cls = getattr(_capi, n)
        mine = C.sizeof(cls) if field == "sizeof" else getattr(cls, field).offset
        assert mine == int(value), "%s.%s: header %s, ctypes %d" % (n, field, value, mine)
        seen += 1
    assert seen > 150
