// Lab-only definitions of the instrumentation points in aero-optical-flow_amd/csrc/aof_lab_hooks.hpp:
// in-kernel s_memrealtime stamps at the phase boundaries of k_coarse (tools/coarse_lab.hip).
#pragma once
__device__ unsigned long long *g_lab_stamps;   // [pairs][8]
#define AOF_LAB_STAMP(slot, k)                                                                      \
    do {                                                                                            \
        if (g_lab_stamps && threadIdx.x == 0) g_lab_stamps[(slot) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
