#pragma once
#include "common.h"
void launch_nhwc4_to_nchw3(const float* in, float* out, int HW, hipStream_t st);   // [HW,4] -> [3,HW]
