// philox.h -- Philox4x32-10 counter-based RNG (Salmon et al., SC'11), host + device.
// Used by the device-side generation of the stretch-move random stream
// (sampler_kernels.h: k_stretch_draw); the same inline code backs the host entry
// point bisip_philox4x32 so the stream can be checked against known-answer vectors.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define BISIP_HD __host__ __device__ __forceinline__
#else
#define BISIP_HD inline
#endif

namespace bisip {

struct Philox4 {
    uint32_t v[4];
};

BISIP_HD Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                               uint32_t k1)
{
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;   // multipliers
    const uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;   // Weyl key increments
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)M0 * c0;
        const uint64_t p1 = (uint64_t)M1 * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    Philox4 o;
    o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}

// 53-bit uniform in [0,1) from two words (the construction NumPy uses for doubles)
BISIP_HD double u53(uint32_t a, uint32_t b)
{
    return (double)(((uint64_t)(a >> 5) << 26) | (uint64_t)(b >> 6)) * (1.0 / 9007199254740992.0);
}

}  // namespace bisip
