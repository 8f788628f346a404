// Step epilogue: region-mask noise composition + CFG + scheduler update + background blend in one launch.
#pragma once
#include "common.h"

struct StepArgs {
    const float* eps;      // [F, HW, 4] (NHWC, 4 channels) UNet outputs of all streams
    const float* masks;    // [R, 4, HW]
    float* lat;            // [4, HW] in/out
    float* lat_ref;        // [4, HW] in/out (may be untouched)
    int HW, R;
    int s_uncond, s_base, s_uref, s_tref;   // stream indices; s_uref < 0 => no reference pair
    int s_region[RT_MAXB];                  // stream of region r (r < R-1); plain mode: R == 0
    float g;
    int plain;             // 1: eps = eps[s_uncond] + g (eps[s_base] - eps[s_uncond]) without masks
    int sched;             // RT_SCHED_EULER / RT_SCHED_PNDM
    int step_ref;          // advance lat_ref too
    float dsigma;          // Euler: sigma_{i+1} - sigma_i
    int pndm_mode;         // 0 first call, 1 second call (counter == 1), 2/3/4 = 2/3/4 stored eps
    float ca, cb;          // PNDM: x_prev = ca * sample - cb * eps'
    float* ets[4];         // PNDM history, ets[0] = slot to write the current eps (if push), ets[1..3] = older
    float* cur_sample;     // PNDM: [2][4*HW]
    int push;
    int blend;             // lat = lat_ref * M[R-1] + lat * (1 - M[R-1]) after the update
    float* noise_pred;     // optional [4, HW]: CFG-combined prediction of the main stream (input of the guidance step)
};

struct IdxList { int v[RT_MAXB]; };
