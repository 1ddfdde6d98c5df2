// Externally built kernels following the reference's device ABI
// (crates/cubecl-cpp/src/hip/signature.rs:28-62): one pointer per buffer binding, then the
// `info` pointer; dynamic LDS is a single `extern __shared__` block.  Loaded through
// mi355_module_load / mi355_launch by tests/test_gpu_runtime.py.
#include <hip/hip_runtime.h>
#include <stdint.h>

struct info_st { uint32_t scalars[2]; uint32_t buffer_len[2]; };

// out[i] = in[i] * scale + bias for i < buffer_len[1]; scalars = {scale, bias} as u32
extern "C" __global__ void __launch_bounds__(1024)
abi_axpb(const uint32_t *const __restrict__ in, uint32_t *const __restrict__ out, const info_st *const __restrict__ info)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < info->buffer_len[1]) out[i] = in[i] * info->scalars[0] + info->scalars[1];
}

// sum_basic (examples/sum_things/src/lib.rs:6-19): every unit sums the whole input sequentially
extern "C" __global__ void __launch_bounds__(1024)
abi_sum_basic(const float *const __restrict__ in, float *const __restrict__ out, const info_st *const __restrict__ info)
{
    float sum = 0.0f;
    for (uint32_t i = 0; i < info->buffer_len[0]; ++i) sum += in[i];
    out[threadIdx.x] = sum;
}

// uses exactly the dynamic LDS it is launched with (runtime_tests/launch.rs:202-224)
extern "C" __global__ void __launch_bounds__(1024)
abi_lds_fill(uint32_t *const __restrict__ out, const info_st *const __restrict__ info)
{
    extern __shared__ uint32_t lds[];
    const uint32_t words = info->scalars[0];
    for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) lds[i] = i;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t acc = 0; for (uint32_t i = 0; i < words; ++i) acc += lds[i]; out[0] = acc; }
}
