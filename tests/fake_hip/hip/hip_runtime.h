// A host-only stand-in for <hip/hip_runtime.h>: just what cubecl_amd/csrc/internal.hpp and pool.cpp use, so that the
// memory pool's logic can be compiled with g++ and tested without a device (tests/test_pool_cpu.py) -- the role the
// reference gives its DummyServer over BytesStorage (crates/cubecl-runtime/tests/dummy/).  TEST INFRASTRUCTURE ONLY:
// nothing in the product includes this directory.
#pragma once
#include <cstddef>
#include <cstdint>

typedef int hipError_t;
enum {
    hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNoDevice = 100, hipErrorInvalidDevice = 101,
    hipErrorNotFound = 500, hipErrorNotReady = 600, hipErrorLaunchOutOfResources = 701, hipErrorLaunchFailure = 719
};
enum { hipEventDefault = 0, hipEventDisableTiming = 2 };

struct fake_hip_stream;
struct fake_hip_event;
typedef fake_hip_stream *hipStream_t;
typedef fake_hip_event *hipEvent_t;

extern "C" {
hipError_t hipSetDevice(int device);
hipError_t hipGetLastError(void);
const char *hipGetErrorString(hipError_t e);
hipError_t hipMalloc(void **ptr, size_t bytes);
hipError_t hipFree(void *ptr);
hipError_t hipDeviceSynchronize(void);
hipError_t hipEventCreateWithFlags(hipEvent_t *event, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t event);
hipError_t hipEventRecord(hipEvent_t event, hipStream_t stream);
hipError_t hipEventQuery(hipEvent_t event);
}
