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

enum { hipStreamNonBlocking = 1, hipHostMallocMapped = 2 };
typedef enum { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 } hipMemcpyKind;
typedef enum { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1 } hipStreamCaptureMode;
typedef enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 } hipFuncAttribute;
struct fake_hip_module;
struct fake_hip_function;
struct fake_hip_graph;
struct fake_hip_graph_exec;
typedef fake_hip_module *hipModule_t;
typedef fake_hip_function *hipFunction_t;
typedef fake_hip_graph *hipGraph_t;
typedef fake_hip_graph_exec *hipGraphExec_t;
struct hipDeviceProp_t {            // the fields runtime.cpp reads
    char name[256];
    char gcnArchName[256];
    size_t totalGlobalMem, sharedMemPerBlock, maxSharedMemoryPerMultiProcessor, textureAlignment, surfaceAlignment;
    int warpSize, maxThreadsPerBlock, maxThreadsDim[3], maxGridSize[3], multiProcessorCount, clockRate, memoryClockRate,
        memoryBusWidth, l2CacheSize;
};

extern "C" {
hipError_t hipGetDeviceCount(int *count);
hipError_t hipGetDeviceProperties(hipDeviceProp_t *prop, int device);
hipError_t hipMemGetInfo(size_t *free_bytes, size_t *total_bytes);
hipError_t hipDeviceCanAccessPeer(int *can, int device, int peer);
hipError_t hipStreamCreateWithFlags(hipStream_t *stream, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t stream);
hipError_t hipStreamSynchronize(hipStream_t stream);
hipError_t hipStreamWaitEvent(hipStream_t stream, hipEvent_t event, unsigned flags);
hipError_t hipEventCreate(hipEvent_t *event);
hipError_t hipEventSynchronize(hipEvent_t event);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t start, hipEvent_t stop);
hipError_t hipHostMalloc(void **ptr, size_t bytes, unsigned flags);
hipError_t hipHostFree(void *ptr);
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t stream);
hipError_t hipMemcpy2DAsync(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height,
                            hipMemcpyKind kind, hipStream_t stream);
hipError_t hipMemcpyPeerAsync(void *dst, int dst_device, const void *src, int src_device, size_t bytes, hipStream_t stream);
hipError_t hipMemsetAsync(void *dst, int value, size_t bytes, hipStream_t stream);
hipError_t hipMemset(void *dst, int value, size_t bytes);
hipError_t hipModuleLoadData(hipModule_t *module, const void *image);
hipError_t hipModuleUnload(hipModule_t module);
hipError_t hipModuleGetFunction(hipFunction_t *function, hipModule_t module, const char *name);
hipError_t hipModuleLaunchKernel(hipFunction_t f, unsigned gx, unsigned gy, unsigned gz, unsigned bx, unsigned by, unsigned bz,
                                 unsigned shared_bytes, hipStream_t stream, void **params, void **extra);
hipError_t hipFuncSetAttribute(const void *func, hipFuncAttribute attr, int value);
hipError_t hipStreamBeginCapture(hipStream_t stream, hipStreamCaptureMode mode);
hipError_t hipStreamEndCapture(hipStream_t stream, hipGraph_t *graph);
hipError_t hipGraphInstantiate(hipGraphExec_t *exec, hipGraph_t graph, void *error_node, char *log, size_t log_bytes);
hipError_t hipGraphLaunch(hipGraphExec_t exec, hipStream_t stream);
hipError_t hipGraphExecDestroy(hipGraphExec_t exec);
hipError_t hipGraphDestroy(hipGraph_t graph);
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
