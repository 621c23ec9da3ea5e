"""Build recipes for the native libraries (in-tree, so the .so files travel with gpurun).

    python -m mitsuba2_amd.build            # product libraries (hipcc, g++)
    python -m mitsuba2_amd.build --oracle   # + the CPU checker under oracle/_build

Like the reference, one build = one variant: everything below exists twice, scalar_rgb (no suffix)
and scalar_spectral (suffix _spectral, compiled with -DMIW_SPECTRAL=1: 4-wavelength Spectrum).
Product:
  mitsuba2_amd/lib/libmiwave[_spectral].so       gfx950 kernels + C ABI (include/miwave.h)        [hipcc]
  mitsuba2_amd/lib/libmiwave_host[_spectral].so  C++17 host classes + ctypes facade               [g++]
Checker (test infrastructure, never loaded by the package):
  oracle/_build/libmiw_oracle[_spectral].so      scalar restatement + CPU wavefront emulator      [g++]
  oracle/_ref/                                   built from the reference's own ext/rgb2spec sources (see build_oracle_ref)

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
        out += [os.path.join(base, f) for f in files if f.endswith((".h", ".hpp", ".hip", ".cpp", ".inl"))]
    return out


def _run(cmd):
    print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


VARIANTS = {"scalar_rgb": ("", []), "scalar_spectral": ("_spectral", ["-DMIW_SPECTRAL=1"])}


def build_device(force=False, variant="scalar_rgb"):
    os.makedirs(LIB, exist_ok=True)
    suffix, defs = VARIANTS[variant]
    out = os.path.join(LIB, "libmiwave%s.so" % suffix)
    src = os.path.join(PKG, "csrc", "miwave.hip")
    deps = _headers(os.path.join(PKG, "csrc")) + [os.path.join(ROOT, "include", "miwave.h")]
    if force or _newer(out, deps):
        _run([HIPCC] + HIP_FLAGS + defs + [src, "-o", out])
    return out


def build_host(force=False, variant="scalar_rgb"):
    os.makedirs(LIB, exist_ok=True)
    suffix, defs = VARIANTS[variant]
    out = os.path.join(LIB, "libmiwave_host%s.so" % suffix)
    src = os.path.join(PKG, "host", "miwave_host.cpp")
    deps = _headers(os.path.join(PKG, "host")) + _headers(os.path.join(PKG, "csrc", "miw")) + \
        [os.path.join(ROOT, "include", "miwave.h"), os.path.join(LIB, "libmiwave%s.so" % suffix)]
    if force or _newer(out, deps):
        _run([CXX] + CXX_FLAGS + defs + [src, "-o", out, "-L" + LIB, "-lmiwave%s" % suffix, "-Wl,-rpath,$ORIGIN"])
    return out


def build_oracle(force=False, variant="scalar_rgb"):
    os.makedirs(ORACLE_BUILD, exist_ok=True)
    suffix, defs = VARIANTS[variant]
    out = os.path.join(ORACLE_BUILD, "libmiw_oracle%s.so" % suffix)
    srcs = [os.path.join(ROOT, "oracle", "miw_oracle.cpp"), os.path.join(ROOT, "oracle", "wavefront_emu.cpp")]
    deps = srcs + _headers(os.path.join(PKG, "csrc")) + [os.path.join(ROOT, "include", "miwave.h")]
    if force or _newer(out, deps):
        _run([CXX] + CXX_FLAGS + defs + srcs + ["-o", out, "-lpthread"])
    return out


REFERENCE = os.environ.get("MIWAVE_REFERENCE", "/root/reference")
ORACLE_REF = os.path.join(ROOT, "oracle", "_ref")


def build_oracle_ref(force=False):
    """oracle/_ref: what can be built of the real reference, from its sources where they lie
    (ext/rgb2spec is self-contained C/C++; the renderer itself needs the empty ext/ submodules):
      rgb2spec_opt        the reference's spectral-upsampling optimiser  (ext/rgb2spec/rgb2spec_opt.cpp)
      srgb.coeff          its output `rgb2spec_opt 64 srgb.coeff` = the reference's data/srgb.coeff
                          (ext/rgb2spec/CMakeLists.txt:47-52); ~50 s single-threaded, built once
      librgb2spec_ref.so  rgb2spec_fetch / rgb2spec_load (ext/rgb2spec/rgb2spec.c), to check the host layer's fetch
    Only present where /root/reference is (this container); the GPU box uses the files that travel with the repo."""
    src_dir = os.path.join(REFERENCE, "ext", "rgb2spec")
    if not os.path.isdir(src_dir):
        return None
    os.makedirs(ORACLE_REF, exist_ok=True)
    opt = os.path.join(ORACLE_REF, "rgb2spec_opt")
    coeff = os.path.join(ORACLE_REF, "srgb.coeff")
    lib = os.path.join(ORACLE_REF, "librgb2spec_ref.so")
    if force or not os.path.exists(opt):
        _run([CXX, "-O2", "-std=c++11", os.path.join(src_dir, "rgb2spec_opt.cpp"), "-o", opt])
    if force or not os.path.exists(coeff):
        _run([opt, "64", coeff])
    if force or not os.path.exists(lib):
        _run(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-I" + src_dir, os.path.join(src_dir, "rgb2spec.c"), "-o", lib, "-lm"])
    return coeff


def build_all(oracle=True, force=False):
    for variant in VARIANTS:
        build_device(force, variant)
        build_host(force, variant)
        if oracle:
            build_oracle(force, variant)
    if oracle:
        build_oracle_ref(False)


if __name__ == "__main__":
    build_all(oracle="--oracle" in sys.argv or "--all" in sys.argv, force="--force" in sys.argv)
