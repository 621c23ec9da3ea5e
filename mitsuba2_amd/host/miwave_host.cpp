// Implementation of the host classes declared in miwave_host.h, plus the C
// facade (`mih_*`) that Python's ctypes binds for tests and the benchmark.
#include "miwave_host.h"

#include <algorithm>
#include <functional>
#include <cmath>
#include <cstring>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <limits>

#include "../csrc/miw/base.h"
#include "../csrc/miw/rng.h"
#include "../csrc/miw/bsdf.h"
#include "../csrc/miw/scene.h"
#include "../csrc/miw/special.h"
#include <cctype>

namespace miwave {

static std::string to_lower(std::string s) { for (char &c : s) c = (char) std::tolower(c); return s; }

[[noreturn]] static void Throw(const std::string &msg) { throw std::runtime_error(msg); }

#include "parts/core.inl"
#include "parts/film_sensor.inl"
#include "parts/bsdfs.inl"
#include "parts/shapes_scene.inl"
#include "parts/integrators.inl"
#include "parts/xml.inl"
#include "parts/facade.inl"
