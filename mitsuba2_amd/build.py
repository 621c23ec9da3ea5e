"""Build recipes for the native libraries (in-tree, so the .so files travel with gpurun).

    python -m mitsuba2_amd.build            # product libraries (hipcc, g++)
    python -m mitsuba2_amd.build --oracle   # + the CPU checker under oracle/_build

Product:
  mitsuba2_amd/lib/libmiwave.so       gfx950 kernels + C ABI (include/miwave.h)        [hipcc]
  mitsuba2_amd/lib/libmiwave_host.so  C++17 host classes + ctypes facade               [g++]
Checker (test infrastructure, never loaded by the package):
  oracle/_build/libmiw_oracle.so      scalar_rgb restatement + CPU wavefront emulator  [g++]

Float flags are part of the parity contract (miw/base.h): no contraction, no
fast-math, correctly rounded div/sqrt (hipcc default), denormals preserved on
both sides (hipcc's default f32 denormal mode on gfx950; MXCSR FTZ/DAZ forced off in
the checker). The reference flushes denormals on the CPU (integrator.cpp:117) for
speed; x86 FTZ/DAZ and gfx950 flush mode disagree on denormal pass-through, so
plain IEEE is the one arithmetic both targets implement identically.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "mitsuba2_amd")
LIB = os.path.join(PKG, "lib")
ORACLE_BUILD = os.path.join(ROOT, "oracle", "_build")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CXX = os.environ.get("CXX", "g++")

HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
             "-fno-gpu-flush-denormals-to-zero", "-fPIC", "-shared"]
CXX_FLAGS = ["-O2", "-std=c++17", "-ffp-contract=off", "-mfma", "-fPIC", "-shared"]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _headers(d):
    out = []
    for base, _, files in os.walk(d):
        out += [os.path.join(base, f) for f in files if f.endswith((".h", ".hpp", ".hip", ".cpp"))]
    return out


def _run(cmd):
    print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build_device(force=False):
    os.makedirs(LIB, exist_ok=True)
    out = os.path.join(LIB, "libmiwave.so")
    src = os.path.join(PKG, "csrc", "miwave.hip")
    deps = _headers(os.path.join(PKG, "csrc")) + [os.path.join(ROOT, "include", "miwave.h")]
    if force or _newer(out, deps):
        _run([HIPCC] + HIP_FLAGS + [src, "-o", out])
    return out


def build_host(force=False):
    os.makedirs(LIB, exist_ok=True)
    out = os.path.join(LIB, "libmiwave_host.so")
    src = os.path.join(PKG, "host", "miwave_host.cpp")
    deps = _headers(os.path.join(PKG, "host")) + _headers(os.path.join(PKG, "csrc", "miw")) + \
        [os.path.join(ROOT, "include", "miwave.h"), os.path.join(LIB, "libmiwave.so")]
    if force or _newer(out, deps):
        _run([CXX] + CXX_FLAGS + [src, "-o", out, "-L" + LIB, "-lmiwave", "-Wl,-rpath,$ORIGIN"])
    return out


def build_oracle(force=False):
    os.makedirs(ORACLE_BUILD, exist_ok=True)
    out = os.path.join(ORACLE_BUILD, "libmiw_oracle.so")
    srcs = [os.path.join(ROOT, "oracle", "miw_oracle.cpp"), os.path.join(ROOT, "oracle", "wavefront_emu.cpp")]
    deps = srcs + _headers(os.path.join(PKG, "csrc")) + [os.path.join(ROOT, "include", "miwave.h")]
    if force or _newer(out, deps):
        _run([CXX] + CXX_FLAGS + srcs + ["-o", out, "-lpthread"])
    return out


def build_all(oracle=True, force=False):
    build_device(force)
    build_host(force)
    if oracle:
        build_oracle(force)


if __name__ == "__main__":
    build_all(oracle="--oracle" in sys.argv or "--all" in sys.argv, force="--force" in sys.argv)
