// fp8.hpp -- OCP FP8 (e4m3fn / e5m2) <-> f32 on the device, in integer arithmetic so that the bits are the
// ones the CPU oracle produces (oracle/oracle.c, same rules): round to nearest even, saturate to +-MAX
// (infinities included), NaN stays NaN, subnormals kept.  Contract: crates/cubecl-common/src/float/fp8/
// fp8_e4m3.rs:77-100 and fp8_e5m2.rs:78-100 of the reference (gfx950's matrix cores read the same OCP encodings;
// MI300's fnuz variants are a different format and are not used anywhere here).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mi355 {

template <int MBITS, int BIAS, uint32_t MAX_CODE>
__device__ __forceinline__ uint8_t f32_to_fp8_bits(float f)
{
    const uint32_t u = __float_as_uint(f);
    const uint32_t sign = (u >> 24) & 0x80u;
    const uint32_t a = u & 0x7FFFFFFFu;
    if (a > 0x7F800000u) return (uint8_t)(sign | 0x7Fu);
    uint32_t code;
    if (a < ((uint32_t)(128 - BIAS) << 23)) {                     // below the smallest normal 2^(1-BIAS)
        // multiples of 2^(1-BIAS-MBITS); the scale is a power of two (exact), rintf rounds to nearest even
        code = (uint32_t)rintf(__uint_as_float(a) * __uint_as_float((uint32_t)(127 + BIAS - 1 + MBITS) << 23));
    } else {
        constexpr int SHIFT = 23 - MBITS;
        const uint32_t r = a + ((1u << (SHIFT - 1)) - 1u) + ((a >> SHIFT) & 1u);
        code = (r >> SHIFT) - ((uint32_t)(127 - BIAS) << MBITS);
    }
    if (code > MAX_CODE) code = MAX_CODE;
    return (uint8_t)(sign | code);
}
__device__ __forceinline__ uint8_t f32_to_e4m3(float f) { return f32_to_fp8_bits<3, 7, 0x7Eu>(f); }
__device__ __forceinline__ uint8_t f32_to_e5m2(float f) { return f32_to_fp8_bits<2, 15, 0x7Bu>(f); }

__device__ __forceinline__ float e4m3_to_f32(uint8_t b)
{
    const uint32_t e = (b >> 3) & 15u, m = b & 7u, sign = (uint32_t)(b & 0x80u) << 24;
    if (e == 15u && m == 7u) return __uint_as_float(sign | 0x7FC00000u);
    if (e == 0u) return __uint_as_float(sign | __float_as_uint((float)m * (1.0f / 512.0f)));
    return __uint_as_float(sign | ((e + 120u) << 23) | (m << 20));
}
__device__ __forceinline__ float e5m2_to_f32(uint8_t b)
{
    // an e5m2 value is the high byte of the binary16 with the same sign / exponent / leading mantissa bits
    const uint16_t hbits = (uint16_t)((uint16_t)b << 8);
    _Float16 h;
    __builtin_memcpy(&h, &hbits, 2);
    return (float)h;
}

}  // namespace mi355
