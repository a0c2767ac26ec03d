// Host build of csrc/pt_math.h behind one C entry point, so that the kernels the device code evaluates can be checked on
// a machine without a GPU (accuracy, and bit equality with any other statement of them).
#include <stdint.h>

#include "pt_math.h"

extern "C" void ptm_eval(int which, uint64_t n, const float *a, const float *b, float *out0, float *out1)
{
    for (uint64_t i = 0; i < n; i++) {
        switch (which) {
        case 0: ptm::sincos_turn(a[i], &out0[i], &out1[i]); break;
        case 1: out0[i] = ptm::atan2_fixed(a[i], b[i]); break;
        case 2: out0[i] = ptm::acos_fixed(a[i]); break;
        default: out0[i] = ptm::pow5(a[i]); break;
        }
    }
}
