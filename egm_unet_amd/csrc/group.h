// Launch groups: INDEPENDENT convolutions recorded between egm_group_begin() and egm_group_end() and launched together -- those that
// run the same kernel instantiation as ONE launch whose workgroups are split between the members by a block-index prefix.
//
// The parallel branches of EdgeEnhancedGRFB (src/EGM-UNet.py:1256-1278) put three convolutions of 16..128 channels side by side at
// every depth; at the 64^2 and 32^2 levels each of them has 32-256 workgroups for 256 CUs (two resident per CU), so three launches
// in a row leave most of the chip idle three times.  Independent branches on forked streams inside a captured graph are no answer on
// this ROCm (2x slower); one launch that carries all three is.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

constexpr int EGM_GROUP_MAX = 4;        // members of one merged launch

struct EgmGroupRec {
    // launches recs[0..n) (n <= EGM_GROUP_MAX, all recorded by the same instantiation) on st
    int (*launch)(const EgmGroupRec* recs, int n, hipStream_t st);
    alignas(16) unsigned char params[384];   // the kernel's parameter struct, copied
    int G;                                    // pixel groups (kernels that take it beside the struct)
    int grid;                                 // workgroups of this member (multiple of 8: XCD alignment of the next member)
    size_t smem;                              // dynamic LDS bytes
};

bool egm_group_recording();
void egm_group_set_recording(bool on);      // pause / resume an open group (a launch whose result is consumed at once)
void egm_group_push(const EgmGroupRec& r);
