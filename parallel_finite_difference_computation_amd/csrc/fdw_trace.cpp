// fdw_trace.cpp -- roctx ranges around the phases of the host loops (forward loop, backward loop, halo exchange, whole shots), so that a
// `rocprofv3 --marker-trace --kernel-trace` timeline shows which launches belong to which phase.  The reference only has a wall-clock
// printf around the whole run (cuda_reference_RTM/src/fd-code.cu:393,535-538).
//
// The marker library is opened with dlopen like RCCL (fdw_comm.cpp): librocprofiler-sdk-roctx (what rocprofv3 listens to), else the older
// libroctx64.  Nothing is loaded -- and a range costs one predictable branch -- unless FDW_ROCTX=1 is set or a marker library is already
// in the process (a profiler put it there), so production runs pay nothing.
#include <dlfcn.h>

#include <cstdlib>
#include <mutex>

#include "fdw_internal.h"

namespace {
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    bool on = false;
};
Roctx* roctx()
{
    static Roctx r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"};
        const char* want = getenv("FDW_ROCTX");
        if (want && atoi(want) == 0) return;                   // FDW_ROCTX=0: never
        void* so = nullptr;
        for (const char* n : names)
            if ((so = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;      // already in the process?
        if (!so && want)
            for (const char* n : names)
                if ((so = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!so) return;
        *(void**)(&r.push) = dlsym(so, "roctxRangePushA");
        *(void**)(&r.pop) = dlsym(so, "roctxRangePop");
        r.on = r.push && r.pop;
    });
    return &r;
}
}  // namespace

fdw_range::fdw_range(const char* name) : active(roctx()->on)
{
    if (active) (void)roctx()->push(name);
}
fdw_range::~fdw_range()
{
    if (active) (void)roctx()->pop();
}
extern "C" int fdw_trace_active(void) { return roctx()->on ? 1 : 0; }
