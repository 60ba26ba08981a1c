// host_rng.cpp -- the stretch move's random stream in numpy.random.RandomState order,
// generated in C for a whole chunk of iterations.
//
// bisip_amd/sampler.py:draw_step is the contract (one uniform double for the choice over the move list, a shuffle
// of the split labels, then per half rand / randint / rand).  This file replays exactly the
// MT19937 consumption of those legacy RandomState methods -- random_sample (two 32-bit words
// per double), masked rejection for bounded integers and for the Fisher-Yates shuffle -- so
// that the stream, and the RandomState left behind, are bit-identical to calling NumPy step
// by step (tests/test_host_logic.py checks it against NumPy itself), at ~1/30 of the cost.
// Logs are NOT taken here: the caller applies numpy.log to the returned zz and u.
#include <cstdint>
#include <vector>

#include "../../include/bisip_hip.h"

namespace {

struct MT {
    uint32_t *key;  // 624 words
    int pos;
    uint32_t out[624];   // tempered words of the current state block, valid from `pos` on
    bool fresh = false;  // out[] matches key[]
    void gen()
    {
        const int N = 624, M = 397;
        const uint32_t A = 0x9908b0dfu, UP = 0x80000000u, LO = 0x7fffffffu;
        int i;
        uint32_t y;
        for (i = 0; i < N - M; i++) {
            y = (key[i] & UP) | (key[i + 1] & LO);
            key[i] = key[i + M] ^ (y >> 1) ^ (-(int32_t)(y & 1) & A);
        }
        for (; i < N - 1; i++) {
            y = (key[i] & UP) | (key[i + 1] & LO);
            key[i] = key[i + (M - N)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & A);
        }
        y = (key[N - 1] & UP) | (key[0] & LO);
        key[N - 1] = key[M - 1] ^ (y >> 1) ^ (-(int32_t)(y & 1) & A);
        pos = 0;
        fresh = false;
    }
    // tempering of the whole block in one vectorisable loop (the state words themselves stay
    // untempered in key[], as NumPy keeps them)
    void temper_block()
    {
        for (int i = 0; i < 624; i++) {
            uint32_t y = key[i];
            y ^= (y >> 11);
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= (y >> 18);
            out[i] = y;
        }
        fresh = true;
    }
    inline uint32_t next32()
    {
        if (pos == 624) gen();
        if (!fresh) temper_block();
        return out[pos++];
    }
    inline double next_double()
    {
        const int32_t a = next32() >> 5, b = next32() >> 6;
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
    static inline uint32_t mask_for(uint32_t max)
    {
        uint32_t mask = max;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        return mask;
    }
    // masked rejection on [0, max], max < 2^32 (random_interval / bounded_masked_uint32)
    inline uint32_t bounded(uint32_t max, uint32_t mask)
    {
        if (max == 0) return 0;
        uint32_t v;
        while ((v = (next32() & mask)) > max) {}
        return v;
    }
};

}  // namespace

extern "C" int bisip_numpy_stretch_stream(uint32_t *mt_key, int32_t *mt_pos, int64_t W, double a,
                                          int64_t n_steps, int32_t *active, int32_t *partner,
                                          double *zz, double *u)
{
    if (!mt_key || !mt_pos || W < 2 || W > 0x7fffffffLL || n_steps < 0 || *mt_pos < 0 || *mt_pos > 624)
        return BISIP_EINVAL;
    if (n_steps == 0) return BISIP_OK;   // nothing to draw: the output arrays may be empty
    if (!active || !partner || !zz || !u) return BISIP_EINVAL;
    MT mt;
    mt.key = mt_key;
    mt.pos = *mt_pos;
    const int64_t nh = (W + 1) / 2;
    std::vector<int32_t> inds(W), half[2];
    half[0].reserve(nh); half[1].reserve(nh);
    for (int64_t k = 0; k < n_steps; ++k) {
        (void)mt.next_double();   // the weighted choice over the move list: one uniform double
        for (int64_t i = 0; i < W; ++i) inds[i] = (int32_t)(i & 1);
        uint32_t smask = MT::mask_for((uint32_t)(W - 1));
        for (int64_t i = W - 1; i >= 1; --i) {  // rng.shuffle(inds)
            if (((uint32_t)i & ((smask >> 1) + 1)) == 0) smask >>= 1;   // i dropped below a power of two
            const uint32_t j = mt.bounded((uint32_t)i, smask);
            const int32_t tmp = inds[i]; inds[i] = inds[j]; inds[j] = tmp;
        }
        half[0].clear(); half[1].clear();
        for (int64_t i = 0; i < W; ++i) half[inds[i]].push_back((int32_t)i);
        for (int h = 0; h < 2; ++h) {
            const std::vector<int32_t> &act = half[h], &comp = half[1 - h];
            const int64_t Ns = (int64_t)act.size(), Nc = (int64_t)comp.size();
            const int64_t off = (k * 2 + h) * nh;
            for (int64_t t = 0; t < Ns; ++t) {      // zz = ((a-1)*rand + 1)**2 / a
                const double v = (a - 1.0) * mt.next_double() + 1.0;
                zz[off + t] = (v * v) / a;
                active[off + t] = act[t];
            }
            const uint32_t pmask = MT::mask_for((uint32_t)(Nc - 1));
            for (int64_t t = 0; t < Ns; ++t)        // randint(Nc, size=Ns)
                partner[off + t] = comp[mt.bounded((uint32_t)(Nc - 1), pmask)];
            for (int64_t t = 0; t < Ns; ++t) u[off + t] = mt.next_double();   // rand(Ns)
            for (int64_t t = Ns; t < nh; ++t) {      // padding slot of the smaller half
                active[off + t] = 0; partner[off + t] = 0; zz[off + t] = 1.0; u[off + t] = 1.0;
            }
        }
    }
    *mt_pos = mt.pos;
    return BISIP_OK;
}
