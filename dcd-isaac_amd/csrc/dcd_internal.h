// dcd_internal.h — shared host-side helpers of libdcd_hip.so (error channel).
#pragma once
#include <stdarg.h>
#include <string>
namespace dcd {
extern thread_local std::string g_last_error;
int fail(const char* fmt, ...);   // records the message for dcd_last_error(), returns 1
}
