// dcd_internal.h — shared host-side helpers of libdcd_hip.so (error channel).
#pragma once
#include <stdarg.h>
#include <string>
namespace dcd {
extern thread_local std::string g_last_error;
int fail(const char* fmt, ...);   // records the message for dcd_last_error(), returns 1
}

// ---- per-kernel timing (dcd_profile_begin / dcd_profile_end) -------------------------------------------------------------
// While a profile is open, every tagged launch carries a HIP start / stop event pair (hipExtLaunchKernelGGL: the kernel's own
// begin -> end timestamps, what rocprofv3 reads, without a tracer attached).  Outside a profile the events are null and the
// launch is the plain one.
#include <hip/hip_runtime_api.h>
namespace dcd {
enum ProfTag { PROF_STEP = 0, PROF_LEVEL = 1, PROF_TAIL = 2, PROF_SCORE = 3, PROF_COUNT = 4, PROF_OTHER = 5, PROF_NTAGS = 6 };
void prof_pair(int tag, hipEvent_t* start, hipEvent_t* stop);   // leaves both null unless a profile is open and has room
}
#define DCD_LAUNCH_TAGGED(tag, kern, grid, block, stream, ...)                                      \
  do {                                                                                              \
    hipEvent_t _e0 = nullptr, _e1 = nullptr;                                                        \
    dcd::prof_pair(tag, &_e0, &_e1);                                                                \
    hipExtLaunchKernelGGL(kern, grid, block, 0, stream, _e0, _e1, 0, __VA_ARGS__);                  \
  } while (0)
