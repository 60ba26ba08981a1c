// comm_rccl.hip -- the sharded stretch move's per-half-step exchange over RCCL (xGMI).
//
// Walkers shard across the GPUs of one node (one process per GPU); every rank keeps the whole
// ensemble and the same random stream, evaluates only its block of the active half, and ONE
// all-gather per half-step rebuilds the state everywhere (the reference's only
// parallel hook is fit(pool=...) -> emcee pool.map, src/bisip/models.py:84,91-94,115).
//
// bisip_stretch_run_sharded_dev enqueues  eval -> ncclAllGather -> apply  for every half-step
// of a chunk on ONE stream with no host round trip: the eval kernel writes this rank's rows
// straight into its slab of the gather buffer (in-place all-gather, no staging copy), the
// buffer is allocated once per context, and the payload per rank is ceil(slots/world) rows of
// (ndim+2) doubles.
//
// RCCL is bound at run time (dlopen by soname, preferring the instance already in the
// process -- PyTorch's -- so that a communicator created by torch.distributed can be handed
// in as is); libbisip_hip.so itself does not link against it.
#include <dlfcn.h>

#include <mutex>

#include <rccl/rccl.h>

#include "host.h"

using namespace bisip;
using namespace bisip::host;

namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;
std::mutex g_rccl_mutex;

int load_rccl()
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);   // contexts of different threads may arrive together
    if (g_rccl.lib) return BISIP_OK;
    void *h = nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so"}) {
        h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);   // the instance already mapped (PyTorch's), if any
        if (h) break;
    }
    if (!h)
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
    if (!h) return fail(BISIP_EUNSUPPORTED, "RCCL not found (librccl.so.1): %s", dlerror());
    Rccl r;
    r.lib = h;
#define BIND(field, sym)                                                          \
    *(void **)(&r.field) = dlsym(h, sym);                                         \
    if (!r.field) return fail(BISIP_EUNSUPPORTED, "RCCL symbol %s not found", sym)
    BIND(GetUniqueId, "ncclGetUniqueId");
    BIND(CommInitRank, "ncclCommInitRank");
    BIND(CommDestroy, "ncclCommDestroy");
    BIND(CommCount, "ncclCommCount");
    BIND(CommUserRank, "ncclCommUserRank");
    BIND(AllGather, "ncclAllGather");
    BIND(GetErrorString, "ncclGetErrorString");
#undef BIND
    g_rccl = r;
    return BISIP_OK;
}

#define RCCL_TRY(expr)                                                                       \
    do {                                                                                     \
        ncclResult_t r_ = (expr);                                                            \
        if (r_ != ncclSuccess)                                                               \
            return fail(BISIP_ERCCL, "%s failed: %s", #expr, g_rccl.GetErrorString(r_));     \
    } while (0)

}  // namespace

extern "C" {

int bisip_rccl_unique_id(void *id128)
{
    if (!id128) return fail(BISIP_EINVAL, "null argument");
    int rc = load_rccl();
    if (rc != BISIP_OK) return rc;
    ncclUniqueId id;
    RCCL_TRY(g_rccl.GetUniqueId(&id));
    static_assert(sizeof(id) == BISIP_RCCL_ID_BYTES, "unique id size");
    std::memcpy(id128, &id, sizeof(id));
    return BISIP_OK;
}

int bisip_rccl_comm_create(void **comm, int world, int rank, const void *id128, int device)
{
    if (!comm || !id128) return fail(BISIP_EINVAL, "null argument");
    *comm = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail(BISIP_EINVAL, "bad world=%d rank=%d", world, rank);
    int rc = load_rccl();
    if (rc != BISIP_OK) return rc;
    HIP_TRY(hipSetDevice(device));
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    RCCL_TRY(g_rccl.CommInitRank(&c, world, id, rank));
    *comm = c;
    return BISIP_OK;
}

int bisip_rccl_comm_destroy(void *comm)
{
    if (!comm) return BISIP_OK;
    int rc = load_rccl();
    if (rc != BISIP_OK) return rc;
    RCCL_TRY(g_rccl.CommDestroy((ncclComm_t)comm));
    return BISIP_OK;
}

// The sharded half-step loop.  comm != nullptr: this process is rank `rank` of `world`; its block of every
// half-step is evaluated into its slab of the gather buffer and one in-place ncclAllGather assembles the
// others'.  comm == nullptr (bisip_stretch_run_sharded_sim_dev): EVERY rank's block is evaluated here, one
// after another, into the slab that rank would have sent -- the same slot ranges, pads and slab offsets,
// no collective -- so that the arithmetic a multi-rank run depends on can be checked on one GPU.
static int run_sharded(bisip_ctx *c, void *comm, int world, int rank, const bisip_stretch_args *first, int64_t W,
                       int64_t n_steps, int64_t thin_by, void *stream)
{
    if (c->E > 1) return fail(BISIP_EUNSUPPORTED, "a batch of spectra shards as whole replicas: no sharded half-step");
    if (thin_by < 1 || n_steps % thin_by) return fail(BISIP_EINVAL, "n_steps=%lld must be a multiple of thin_by=%lld", (long long)n_steps, (long long)thin_by);
    if (W < 2 || n_steps < 0) return fail(BISIP_EINVAL, "bad W=%lld or n_steps=%lld", (long long)W, (long long)n_steps);
    if (!first->coords || !first->logp || !first->active || !first->partner || !first->zz ||
        !first->factor || !first->logu || !first->status)
        return fail(BISIP_EINVAL, "null buffer");
    if (world < 1 || rank < 0 || rank >= world) return fail(BISIP_EINVAL, "bad world=%d rank=%d", world, rank);
    HIP_TRY(hipSetDevice(c->device));
    const int64_t nh = (W + 1) / 2;
    const int64_t row = c->ndim + 2;                     // position, log-prob, accepted
    const int64_t pad_max = (nh + world - 1) / world;    // rows per rank slab (half 0 is the larger)
    const size_t need = (size_t)world * pad_max * row * sizeof(double);
    if (c->gather_bytes < need) {                        // once per context (and per larger W)
        if (c->d_gather) { HIP_TRY(hipStreamSynchronize((hipStream_t)stream)); (void)hipFree(c->d_gather); c->d_gather = nullptr; c->gather_bytes = 0; }
        hipError_t e = hipMalloc((void **)&c->d_gather, need);
        if (e != hipSuccess) return fail(BISIP_ENOMEM, "hipMalloc(%zu) failed: %s", need, hipGetErrorString(e));
        c->gather_bytes = need;
    }
    bisip_stretch_args u = *first;
    u.world = world;
    int rc = BISIP_OK;
    for (int64_t k = 0; k < n_steps; ++k) {
        for (int h = 0; h < 2; ++h) {
            const int64_t off = (k * 2 + h) * nh;
            u.active = first->active + off; u.partner = first->partner + off;
            u.zz = first->zz + off; u.factor = first->factor + off; u.logu = first->logu + off;
            const int64_t m = h ? W / 2 : nh;
            u.n_slots = m;
            const bool store = ((k + 1) % thin_by) == 0;
            const int64_t srow = k / thin_by;
            u.chain_row = (store && first->chain_row) ? first->chain_row + srow * W * c->ndim : nullptr;
            u.logp_row = (store && first->logp_row) ? first->logp_row + srow * W : nullptr;
            // a rank's block of the active half (dist.py:shard_range): the first m % world ranks
            // own one slot more
            const int64_t base = m / world, extra = m % world;
            const int64_t pad = (m + world - 1) / world;
            u.pad = pad;
            for (int r = comm ? rank : 0; r < (comm ? rank + 1 : world); ++r) {
                u.slot_lo = r * base + (r < extra ? r : extra);
                u.slot_hi = u.slot_lo + base + (r < extra ? 1 : 0);
                u.block = c->d_gather + r * pad * row;
                if (u.slot_hi > u.slot_lo) {
                    const StretchArgs a = to_device_args(&u);
                    rc = dispatch_stretch(c, StretchWork{STRETCH_EVAL, &a, nullptr}, 0, (hipStream_t)stream);
                    if (rc != BISIP_OK) return rc;
                }
            }
            if (comm) {
                // in place: the send buffer IS this rank's slab of the receive buffer
                double *mine = c->d_gather + rank * pad * row;
                RCCL_TRY(g_rccl.AllGather(mine, c->d_gather, (size_t)(pad * row), ncclDouble, (ncclComm_t)comm,
                                          (hipStream_t)stream));
            }
            u.block = c->d_gather;
            rc = dispatch_apply(c, to_device_args(&u), (hipStream_t)stream);
            if (rc != BISIP_OK) return rc;
        }
    }
    return BISIP_OK;
}

int bisip_stretch_run_sharded_dev(bisip_ctx *c, void *comm, const bisip_stretch_args *first, int64_t W,
                                  int64_t n_steps, int64_t thin_by, void *stream)
{
    if (!c || !comm || !first) return fail(BISIP_EINVAL, "null argument");
    int rc = load_rccl();
    if (rc != BISIP_OK) return rc;
    int world = 0, rank = 0;
    RCCL_TRY(g_rccl.CommCount((ncclComm_t)comm, &world));
    RCCL_TRY(g_rccl.CommUserRank((ncclComm_t)comm, &rank));
    if (world < 1 || rank < 0 || rank >= world) return fail(BISIP_EINVAL, "communicator reports world=%d rank=%d", world, rank);
    return run_sharded(c, comm, world, rank, first, W, n_steps, thin_by, stream);
}

int bisip_stretch_run_sharded_sim_dev(bisip_ctx *c, int world, const bisip_stretch_args *first, int64_t W,
                                      int64_t n_steps, int64_t thin_by, void *stream)
{
    if (!c || !first) return fail(BISIP_EINVAL, "null argument");
    if (world < 1 || world > 4096) return fail(BISIP_EINVAL, "world=%d out of [1, 4096]", world);
    return run_sharded(c, nullptr, world, 0, first, W, n_steps, thin_by, stream);
}

}  // extern "C"
