// stretch_launch.h -- launch helpers shared by the two stretch-move dispatch units
// (dispatch_stretch.hip: single spectrum; dispatch_stretch_batch.hip: batch of spectra).
#pragma once
#include "host.h"

#include <cstdlib>

namespace bisip {
namespace host {

// Persistent kernel: workgroup shape and LDS plan (see k_stretch_persistent).
template <class LP>
int launch_persistent(const PersistArgs &p0, const LP &lp, hipStream_t st)
{
    PersistArgs p = p0;
    const long long nh = (p.W + 1) / 2;
    p.lanes_per_ens = (int)(((nh * LP::L + 63) / 64) * 64);          // <= 512: stretch_lanes()
    const size_t state_bytes = (size_t)p.W * (LP::NDIM + 1) * sizeof(double);
    size_t rec_bytes = 0;
    if constexpr (LP::CAN_STAGE) {
        const bool off = std::getenv("BISIP_NO_LDS_STAGING") != nullptr;   // A/B runs (read per call)
        rec_bytes = off ? 0 : (((size_t)lp.n_freq() * LP::REC_DOUBLES + 1) & ~(size_t)1) * sizeof(double);
        if (state_bytes + rec_bytes + 16 > 65536) rec_bytes = 0;                  // does not fit: scalar-cache path
    }
    p.rec_stride = (long long)(rec_bytes / sizeof(double));
    // ensembles per workgroup: up to 256 lanes per workgroup when there are many ensembles
    long long epw = 1;
    if (p.E > 1) {
        epw = 256 / p.lanes_per_ens;                     // (and so never above the 512-lane bound)
        if (epw < 1) epw = 1;
        while (epw > 1 && epw * (state_bytes + rec_bytes) + 16 > 65536) --epw;
        if (epw > p.E) epw = p.E;
    }
    p.epw = (int)epw;
    const unsigned threads = (unsigned)(epw * p.lanes_per_ens);
    const unsigned grid = (unsigned)((p.E + epw - 1) / epw);
    const size_t lds = ((size_t)epw * state_bytes + 15) / 16 * 16 + (size_t)epw * rec_bytes;
    if constexpr (LP::CAN_STAGE) {
        if (rec_bytes) {
            hipLaunchKernelGGL((k_stretch_persistent<LP, true>), dim3(grid), dim3(threads), lds, st, p, lp);
            HIP_TRY(hipGetLastError());
            return BISIP_OK;
        }
    }
    hipLaunchKernelGGL((k_stretch_persistent<LP, false>), dim3(grid), dim3(threads), lds, st, p, lp);
    HIP_TRY(hipGetLastError());
    return BISIP_OK;
}

// One ensemble over several workgroups (k_stretch_group): G workgroups of 256 lanes (64 for one-lane functors) hold the
// larger half's slots, LP::L lanes each; the grid is 8 G (the group is its every eighth workgroup: see the kernel) or G.
constexpr int GROUP_MAX_WORKGROUPS = 256;
template <class LP>
int launch_group(const PersistArgs &p0, const LP &lp, hipStream_t st)
{
    if constexpr (LP::NDIM >= GROUP_ROW || !SingleSpectrum<LP>::value) {
        return fail(BISIP_EUNSUPPORTED, "the multi-workgroup sampler takes one spectrum and keeps a walker in one 64-byte row (ndim=%d)", LP::NDIM);
    } else {
        PersistArgs p = p0;
        const long long nh = (p.W + 1) / 2;
        constexpr int BLK = LP::L == 1 ? 64 : GROUP_BLK;      // (see the kernel: one-lane slots spread over more compute units)
        p.G = (int)((nh * LP::L + BLK - 1) / BLK);
        if (p.G > GROUP_MAX_WORKGROUPS) return fail(BISIP_EUNSUPPORTED, "ensemble of %lld walkers: more than %d workgroups", p.W, GROUP_MAX_WORKGROUPS);
        size_t rec_bytes = 0;
        if constexpr (LP::CAN_STAGE) {
            rec_bytes = (((size_t)lp.n_freq() * LP::REC_DOUBLES + 1) & ~(size_t)1) * sizeof(double);
            if (rec_bytes > 60000 || std::getenv("BISIP_NO_LDS_STAGING") != nullptr) rec_bytes = 0;
        }
        // up to 32 workgroups sit on ONE XCD (every eighth workgroup of the grid); more -- one-lane functors beyond 4,096
        // walkers -- take every workgroup of the grid, on all XCDs, the placement-independent protocol and the barrier
        // in two levels: two workgroups per compute unit of one XCD are slower than a launch per half-step (7.8 us at
        // 8,192 walkers), 64 compute units anywhere are not (3.7 against 4.6)
        p.spread = p.G > 32 ? 1 : 0;
        const dim3 grid((unsigned)(p.spread ? p.G : 8 * p.G)), block(BLK);
        if constexpr (LP::CAN_STAGE) {
            if (rec_bytes) {
                hipLaunchKernelGGL((k_stretch_group<LP, true, BLK>), grid, block, rec_bytes, st, p, lp);
                HIP_TRY(hipGetLastError());
                return BISIP_OK;
            }
        }
        hipLaunchKernelGGL((k_stretch_group<LP, false, BLK>), grid, block, 0, st, p, lp);
        HIP_TRY(hipGetLastError());
        return BISIP_OK;
    }
}

template <class LP>
int launch_stretch(const StretchWork &work, const LP &lp, hipStream_t st)
{
    const StretchKind kind = work.kind;
    if constexpr (LP::L == 8) {       // eight lanes per slot exist for the multi-workgroup sampler alone
        if (kind != STRETCH_GROUP) return fail(BISIP_EUNSUPPORTED, "internal: eight lanes per slot outside the multi-workgroup sampler");
        return launch_group(*work.persist, lp, st);
    } else {
    if (kind == STRETCH_PERSIST) {
        if constexpr (MayPersist<LP>::value) return launch_persistent(*work.persist, lp, st);
        else return fail(BISIP_EUNSUPPORTED, "internal: a persistent launch for a functor whose waves straddle spectra");
    }
    if (kind == STRETCH_GROUP) return launch_group(*work.persist, lp, st);
    const StretchArgs &a = *work.half;
    if (kind == STRETCH_HALF) {
        // a launch that fills the chip (>= one wave per SIMD) goes out as four-wave workgroups, one
        // per CU; smaller ones as single waves so that they spread over as many CUs as possible
        if (a.packed) {
            // the chunk's state is packed (bisip_stretch_run_dev: a single spectrum, one lane per slot, ndim <= 7)
            if constexpr (LP::L == 1 && LP::NDIM < PACKED_ROW && SingleSpectrum<LP>::value) {
                const unsigned grid = (unsigned)((a.n_slots + 255) / 256);
                hipLaunchKernelGGL((k_stretch_half_packed<LP, 256>), dim3(grid), dim3(256), 0, st, a, lp);
            } else {
                return fail(BISIP_EUNSUPPORTED, "internal: a packed state for a kernel that has no packed form");
            }
        } else if (a.n_slots * LP::L >= 65536) {
            const unsigned grid = (unsigned)((a.n_slots * LP::L + 255) / 256);
            hipLaunchKernelGGL((k_stretch_half<LP, 256>), dim3(grid), dim3(256), 0, st, a, lp);
        } else {
            const unsigned grid = (unsigned)((a.n_slots * LP::L + 63) / 64);
            hipLaunchKernelGGL((k_stretch_half<LP, 64>), dim3(grid), dim3(64), 0, st, a, lp);
        }
    } else {
        // (the sharded half-step: a batch of spectra shards as whole replicas and never comes here)
        if constexpr (SingleSpectrum<LP>::value) {
            const unsigned grid = (unsigned)(((a.slot_hi - a.slot_lo) * LP::L + 63) / 64);
            hipLaunchKernelGGL((k_stretch_eval<LP>), dim3(grid), dim3(64), 0, st, a, lp);
        } else {
            return fail(BISIP_EUNSUPPORTED, "the sharded half-step takes a single-spectrum context");
        }
    }
    HIP_TRY(hipGetLastError());
    return BISIP_OK;
    }
}

// lanes per slot of a stretch dispatch: as many as lanes_per_walker() grants for the number of
// slots evaluated at once (all ensembles' for the persistent kernel, whose workgroups run
// concurrently), capped there by the 512-lane workgroup that holds one ensemble's half.
inline int stretch_lanes(const StretchWork &w)
{
    // tuning knob for benchmarks/: BISIP_STRETCH_LANES=1|2|4 overrides the rule below (the value
    // never changes a result -- logprob_row is bit-identical for every L -- only the wave count)
    if (const char *env = std::getenv("BISIP_STRETCH_LANES")) {
        const int v = std::atoi(env);
        if (w.kind == STRETCH_GROUP && (v == 1 || v == 2 || v == 4 || v == 8)) return v;
        if (v == 1 || v == 2 || v == 4) {
            if (w.kind != STRETCH_PERSIST) return v;
            const long long nh = (w.persist->W + 1) / 2;
            return nh * v <= 512 ? v : (nh * 2 <= 512 ? 2 : 1);
        }
    }
    if (w.kind == STRETCH_GROUP) {
        // One evaluation at one wave per SIMD is what a half-step of this kernel waits for (2.15 of 4.4 us at cfg2 with
        // four lanes per slot): as many lanes per slot -- up to eight, half a DPP row -- as keep the group within the
        // 32 compute units of one XCD, one workgroup each.  (Measured with 64 workgroups, two per compute unit: cfg2
        // with eight lanes 6.1 us against 4.4 with four; 2,048 walkers: eight lanes 3.7 us, four 3.85.)
        const long long nh = (w.persist->W + 1) / 2;
        for (int lanes = 8; lanes > 1; lanes /= 2)
            if (nh * lanes <= 32LL * GROUP_BLK) return lanes;
        return 1;
    }
    if (w.kind == STRETCH_PERSIST) {
        const long long nh = (w.persist->W + 1) / 2;
        const int fit = nh * 4 <= 512 ? 4 : (nh * 2 <= 512 ? 2 : 1);
        const int want = lanes_per_walker(nh * w.persist->E);
        return want < fit ? want : fit;
    }
    if (w.kind == STRETCH_HALF && w.half->packed) return 1;
    return lanes_per_walker(w.kind == STRETCH_HALF ? w.half->n_slots : w.half->slot_hi - w.half->slot_lo);
}

}  // namespace host
}  // namespace bisip
