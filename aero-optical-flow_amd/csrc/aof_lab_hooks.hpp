// Instrumentation points of the product kernels.  The product build compiles every one of them to
// nothing.  A lab tool under tools/ that wants in-kernel measurements builds the kernel source with
// -DAOF_LAB_HOOKS='"<its hooks header under tools/>"' and supplies its own definitions there
// (tools/coarse_lab.hip with tools/coarse_lab_hooks.hpp): no measurement code lives in this directory.
#pragma once

#ifdef AOF_LAB_HOOKS
#include AOF_LAB_HOOKS
#else
#define AOF_LAB_STAMP(slot, k) do { } while (0)   // phase boundary k of work item `slot`
#endif
