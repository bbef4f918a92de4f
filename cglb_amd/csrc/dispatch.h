// Compile-time dispatch over element type, kernel kind and padded input dimension.
#pragma once
#include "cglb_internal.h"
#include "devmath.h"

// BODY sees: typename T, constexpr int KIND, constexpr int DP
#define CGLB_DISPATCH_DP(dp, ...)                                           \
    switch (dp) {                                                           \
        case 1: { constexpr int DP = 1; __VA_ARGS__; } break;               \
        case 2: { constexpr int DP = 2; __VA_ARGS__; } break;               \
        case 3: { constexpr int DP = 3; __VA_ARGS__; } break;               \
        case 4: { constexpr int DP = 4; __VA_ARGS__; } break;               \
        case 6: { constexpr int DP = 6; __VA_ARGS__; } break;               \
        case 8: { constexpr int DP = 8; __VA_ARGS__; } break;               \
        case 10: { constexpr int DP = 10; __VA_ARGS__; } break;             \
        case 12: { constexpr int DP = 12; __VA_ARGS__; } break;             \
        case 16: { constexpr int DP = 16; __VA_ARGS__; } break;             \
        case 20: { constexpr int DP = 20; __VA_ARGS__; } break;             \
        case 24: { constexpr int DP = 24; __VA_ARGS__; } break;             \
        case 28: { constexpr int DP = 28; __VA_ARGS__; } break;             \
        case 32: { constexpr int DP = 32; __VA_ARGS__; } break;             \
        default: return cglb_fail(c, CGLB_ERR_BAD_ARG, "unsupported padded dimension"); \
    }

#define CGLB_DISPATCH_KIND(kind, ...)                                       \
    if ((kind) == CGLB_RBF) { constexpr int KIND = CGLB_RBF; __VA_ARGS__; } \
    else { constexpr int KIND = CGLB_MATERN32; __VA_ARGS__; }

#define CGLB_DISPATCH_T(dtype, ...)                                         \
    if ((dtype) == CGLB_F64) { using T = double; __VA_ARGS__; }             \
    else { using T = float; __VA_ARGS__; }

// BODY additionally sees constexpr int PREC (devmath.h precision level).  Use inside a function template on T: fp32 has a hardware
// exp2 / sqrt and builds one instance only.
#define CGLB_DISPATCH_PREC(c, ...)                                                                                   \
    if constexpr (sizeof(T) == 4) { constexpr int PREC = CGLB_PREC_EXACT; __VA_ARGS__; }                             \
    else if ((c)->precision == CGLB_PREC_EXACT) { constexpr int PREC = CGLB_PREC_EXACT; __VA_ARGS__; }               \
    else if ((c)->precision == CGLB_PREC_LOW) { constexpr int PREC = CGLB_PREC_LOW; __VA_ARGS__; }                   \
    else { constexpr int PREC = CGLB_PREC_FAST; __VA_ARGS__; }

#define CGLB_DISPATCH_ALL(c, ...) \
    CGLB_DISPATCH_T((c)->dtype, CGLB_DISPATCH_KIND((c)->kind, CGLB_DISPATCH_DP((c)->Dp, __VA_ARGS__)))

#define CGLB_LAUNCH_CHECK(c)                                                     \
    do {                                                                         \
        hipError_t _e = hipGetLastError();                                       \
        if (_e != hipSuccess) {                                                  \
            (c)->err = std::string("kernel launch: ") + hipGetErrorString(_e);   \
            return CGLB_ERR_HIP;                                                 \
        }                                                                        \
    } while (0)

struct ScaleParams {
    double center[CGLB_MAX_D_NARROW];
    double scale[CGLB_MAX_D_NARROW];  // kscale / l_d
};
