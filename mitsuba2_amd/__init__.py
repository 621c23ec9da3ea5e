"""mitsuba2_amd — MI355X-native wavefront path-tracing core behind Mitsuba 2's
path-integrator plugin surface.

Layers: include/miwave.h (C ABI) <- csrc/ (gfx950 kernels, libmiwave.so)
        <- host/ (C++17 plugin-shaped classes, libmiwave_host.so)
        <- api.py / scenes.py (ctypes plumbing for tests and bench).
Importing the package does not load native code; the first use of `api` does,
and raises if the libraries are missing (there is no CPU fallback).
"""
__version__ = "0.1.0"
