// fdw_internal.h -- what the translation units of libfdwave.so share beyond the public header (include/fdwave.h).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

#include "fdwave.h"

// sets fdw_last_error() of the calling thread and returns `code`
int fdw_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

constexpr int FDW_COMM_MAX_FIELDS = 8;

// Halo exchange of `nfields` fields with the two neighbouring ranks, enqueued on `stream` (fdw_comm.cpp)
int fdw_comm_exchange(fdw_comm* c, int nfields, float* const* fields, size_t send_lo, size_t recv_lo, size_t send_hi, size_t recv_hi,
                      size_t count, hipStream_t stream);

// A roctx range for the lifetime of the object (fdw_trace.cpp): FDW_RANGE("forward loop"); no-op unless a marker library is in the process
struct fdw_range {
    explicit fdw_range(const char* name);
    ~fdw_range();
    fdw_range(const fdw_range&) = delete;
    fdw_range& operator=(const fdw_range&) = delete;
    bool active;
};
#define FDW_RANGE_CAT2(a, b) a##b
#define FDW_RANGE_CAT(a, b) FDW_RANGE_CAT2(a, b)
#define FDW_RANGE(name) fdw_range FDW_RANGE_CAT(fdw_range_, __LINE__)(name)
