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
#include <cstddef>
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
    // the tempered words left in the current block (at least one: refills when the block is used up)
    inline int available()
    {
        if (pos == 624) gen();
        if (!fresh) temper_block();
        return 624 - pos;
    }
    // Fisher-Yates draws for i = hi, hi-1, ..., lo (all under ONE mask): j[i] = the accepted word for bound i.
    // A rejected word costs a mispredicted branch in the obvious loop (a quarter to a half of all words are
    // rejected); here every word is stored at j[i] and i moves on only when it was acceptable -- no branch on data.
    inline void shuffle_draws(int64_t hi, int64_t lo, uint32_t mask, uint32_t *j)
    {
        int64_t i = hi;
        while (i >= lo) {
            const int n = available();
            int p = pos;
            const int end = pos + n;
            while (p < end && i >= lo) {
                const uint32_t v = out[p++] & mask;
                j[i] = v;
                i -= (int64_t)(v <= (uint32_t)i);
            }
            pos = p;
        }
    }
    // `count` masked-rejection draws on [0, max] under one mask into dst[0..count)
    inline void bounded_run(uint32_t max, uint32_t mask, int64_t count, uint32_t *dst)
    {
        if (max == 0) { for (int64_t t = 0; t < count; ++t) dst[t] = 0; return; }
        int64_t t = 0;
        while (t < count) {
            const int n = available();
            int p = pos;
            const int end = pos + n;
            while (p < end && t < count) {
                const uint32_t v = out[p++] & mask;
                dst[t] = v;                       // (dst has one spare element: a rejected last word lands there)
                t += (int64_t)(v <= max);
            }
            pos = p;
        }
    }
    // `count` doubles (random_sample: two words each, which may straddle two blocks)
    inline void doubles_run(int64_t count, double *dst)
    {
        int64_t t = 0;
        while (t < count) {
            const int n = available();
            const int64_t pairs = n / 2 < count - t ? n / 2 : count - t;
            const uint32_t *w = out + pos;
            for (int64_t q = 0; q < pairs; ++q) {
                const int32_t hi = (int32_t)(w[2 * q] >> 5), lo = (int32_t)(w[2 * q + 1] >> 6);
                dst[t + q] = (hi * 67108864.0 + lo) / 9007199254740992.0;
            }
            pos += (int)(2 * pairs);
            t += pairs;
            if (t < count && 624 - pos == 1) dst[t++] = next_double();     // the pair that straddles the refill
        }
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
    std::vector<uint32_t> draws((std::size_t)W + 1);
    std::vector<double> tmp((std::size_t)nh);
    half[0].resize(nh); half[1].resize(nh);
    for (int64_t k = 0; k < n_steps; ++k) {
        (void)mt.next_double();   // the weighted choice over the move list: one uniform double
        for (int64_t i = 0; i < W; ++i) inds[i] = (int32_t)(i & 1);
        // rng.shuffle(inds): bound i, mask = the smallest 2^m - 1 >= i -- runs of i under one mask, from the top
        for (int64_t hi = W - 1; hi >= 1;) {
            const uint32_t smask = MT::mask_for((uint32_t)hi);
            const int64_t lo = (int64_t)(smask >> 1) + 1;          // the last i that still needs this mask
            mt.shuffle_draws(hi, lo, smask, draws.data());
            for (int64_t i = hi; i >= lo; --i) {
                const uint32_t j = draws[i];
                const int32_t t = inds[i]; inds[i] = inds[j]; inds[j] = t;
            }
            hi = lo - 1;
        }
        int64_t cnt[2] = {0, 0};
        for (int64_t i = 0; i < W; ++i) { const int32_t h = inds[i]; half[h][cnt[h]++] = (int32_t)i; }
        for (int h = 0; h < 2; ++h) {
            const std::vector<int32_t> &act = half[h], &comp = half[1 - h];
            const int64_t Ns = cnt[h], Nc = cnt[1 - h];
            const int64_t off = (k * 2 + h) * nh;
            mt.doubles_run(Ns, tmp.data());
            for (int64_t t = 0; t < Ns; ++t) {      // zz = ((a-1)*rand + 1)**2 / a
                const double v = (a - 1.0) * tmp[t] + 1.0;
                zz[off + t] = (v * v) / a;
                active[off + t] = act[t];
            }
            mt.bounded_run((uint32_t)(Nc - 1), MT::mask_for((uint32_t)(Nc - 1)), Ns, draws.data());   // randint(Nc, size=Ns)
            for (int64_t t = 0; t < Ns; ++t) partner[off + t] = comp[draws[t]];
            mt.doubles_run(Ns, u + off);             // rand(Ns)
            for (int64_t t = Ns; t < nh; ++t) {      // padding slot of the smaller half
                active[off + t] = 0; partner[off + t] = 0; zz[off + t] = 1.0; u[off + t] = 1.0;
            }
        }
    }
    *mt_pos = mt.pos;
    return BISIP_OK;
}
