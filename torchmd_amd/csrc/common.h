// Shared host/device helpers for the tmdhip library (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/tmdhip.h"

namespace tmd {

// ---------------------------------------------------------------------------------------------
// error plumbing: no C++ exception crosses the C ABI
// ---------------------------------------------------------------------------------------------
std::string &last_error();
int fail(const std::string &msg);

#define TMD_HIP(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess)                                                                          \
      return ::tmd::fail(std::string(#expr) + ": " + hipGetErrorString(_e) + " (" + __FILE__ +    \
                         ":" + std::to_string(__LINE__) + ")");                                    \
  } while (0)

#define TMD_TRY(expr)                                                                              \
  do {                                                                                             \
    int _r = (expr);                                                                               \
    if (_r != 0) return _r;                                                                        \
  } while (0)

// ---------------------------------------------------------------------------------------------
// real-type traits
// ---------------------------------------------------------------------------------------------
template <typename R>
struct Vec;
template <>
struct Vec<float> {
  using T2 = float2;
  using T4 = float4;
};
template <>
struct Vec<double> {
  using T2 = double2;
  using T4 = double4;
};

constexpr int kWave = 64;  // CDNA wavefront

// Coulomb constant in kcal/mol*A/e^2 — value the reference derives from scipy.constants
// (torchmd/forces.py:375-378).
constexpr double kElecFactor = 332.06371307417066;

}  // namespace tmd
