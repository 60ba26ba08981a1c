/*
 * bisip_hip.h -- C ABI of libbisip_hip.so, the MI355X (gfx950) implementation of
 * BISIP's ensemble-MCMC log-probability hot path.
 *
 * This is the drop-in boundary.  Every entry point replaces a
 * piece of the reference's per-walker Python/Cython path with its batched
 * (walker-vectorised) form; citations are relative to /root/reference:
 *
 *   bisip_ctx_create      the walker-independent operands the reference binds into
 *                         emcee's `args` tuple and the model's precompute:
 *                         src/bisip/models.py:108-109 (forward, bounds, w, zn, zn_err),
 *                         src/bisip/models.py:200-213 (taus, log_taus, c_exp),
 *                         src/bisip/utils.py:138-144 (zn, zn_err, w layout)
 *   bisip_ctx_set_bounds  `model.params.update(...)` between fits; param_bounds is
 *                         re-read at fit() time: src/bisip/models.py:176-179, 102-109
 *   bisip_logprob[_dev]   Inversion._log_probability for W rows at once:
 *                         src/bisip/models.py:59-76 fused with the forward models of
 *                         src/bisip/cython_funcs.pyx:33-108
 *   bisip_forward[_dev]   Model.forward(theta, w) for W rows at once:
 *                         src/bisip/models.py:217-229, 256-271, 295-305, 335-349
 *                         (the loop of src/bisip/utils.py:33-34)
 *
 * Conventions: plain pointers and sizes, no framework types.  All arrays are
 * C-contiguous IEEE binary64.  theta is (W, ndim) row-major -- the layout emcee
 * hands to log_prob_fn -- and row i of the input produces element i of the
 * output.  Functions return 0 on success and a negative BISIP_E* code on failure;
 * bisip_last_error() then describes the failure (thread-local string).
 * A context is bound to one device and is not thread-safe.
 */
#ifndef BISIP_HIP_H
#define BISIP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: entry points added since 1 (all additions, nothing changed or removed): loglike_z, forward_percentiles,
 * column / grouped / columns percentiles, forward_spectrum(s) / forward_columns, stretch_run_sharded + rccl_*, ctx_set_spectrum_offset,
 * ctx_reduced_check, polydecomp_reduced_estimates, read_tables; BISIP_VARIANT_REDUCED_COMP, BISIP_ERCCL.
 * 3: clock_probe_dev, ctx_reduced_guard, polydecomp_reduced_reference, stretch_run_sharded_sim_dev (additions only).
 * 4: chain_shell_rows_dev (+ _workspace), ctx_reduced_guard_rows, ensemble_gram_dev (+ _workspace),
 *    fp64_stream_probe_dev (+ _lanes), stretch_run_philox_dev (+ stretch_philox_inline) (additions only). */
#define BISIP_ABI_VERSION 4

/* model_id -- the four reference model classes (src/bisip/models.py:182,232,274,308) */
#define BISIP_MODEL_POLYDECOMP 0 /* PolynomialDecomposition -> Decomp_cyth  */
#define BISIP_MODEL_COLECOLE   1 /* PeltonColeCole          -> ColeCole_cyth */
#define BISIP_MODEL_DIAS2000   2 /* Dias2000                -> Dias2000_cyth */
#define BISIP_MODEL_SHIN2015   3 /* Shin2015                -> Shin2015_cyth */

/* kernel formulation for PolynomialDecomposition (other models have one) */
#define BISIP_VARIANT_AUTO      0 /* fastest formulation that holds the parity tolerance: REDUCED, else
                                     REDUCED_COMP, else COLLAPSED (see bisip_ctx_reduced_error) */
#define BISIP_VARIANT_FAITHFUL  1 /* sum_k M_k*K[j,k], the reference's loop structure     */
#define BISIP_VARIANT_COLLAPSED 2 /* Z_j = R0*(1 - sum_p a_p*G[j,p])                       */
#define BISIP_VARIANT_REDUCED   3 /* QR-reduced chi^2: (P+2)x(P+2) triangular form         */
#define BISIP_VARIANT_WAVE      4 /* collapsed operands, one wave per walker, LDS-staged
                                     spectrum, wavefront-shuffle reduction (N <= 64)        */
#define BISIP_VARIANT_REDUCED_COMP 5 /* the QR-reduced form in compensated (double-double row
                                     sums) arithmetic: for nearly collinear designs        */

/* status codes */
#define BISIP_OK          0
#define BISIP_EINVAL     -1 /* bad argument (shape, null pointer, unsupported size) */
#define BISIP_EHIP       -2 /* a HIP runtime call failed                             */
#define BISIP_ENOMEM     -3
#define BISIP_EUNSUPPORTED -4
#define BISIP_ERCCL     -5 /* an RCCL call failed                                   */

#define BISIP_MAX_NDIM 16
#define BISIP_MAX_MODES 5
#define BISIP_MAX_POLY_DEG 10

typedef struct bisip_ctx bisip_ctx;

/* Model-specific description.  Unused fields are ignored. */
typedef struct bisip_model_desc {
    int n_modes;            /* COLECOLE: number of Cole-Cole modes D (ndim = 1+3D)        */
    int poly_deg;           /* POLYDECOMP: P (ndim = P+2)                                 */
    int n_taus;             /* POLYDECOMP: S, length of the relaxation-time grid           */
    double c_exp;           /* POLYDECOMP: fixed Cole-Cole exponent (1 Debye, 0.5 Warburg) */
    const double *taus;     /* POLYDECOMP: (S,)       = 10**log_tau                        */
    const double *log_taus; /* POLYDECOMP: (P+1, S)   = log_tau**i, row-major              */
} bisip_model_desc;

/* Create a context on `device`.  w (N,), zn (2,N) [row 0 real, row 1 imaginary],
 * zn_err (2,N), lo/hi (ndim,) are copied; the caller keeps ownership. */
int bisip_ctx_create(bisip_ctx **out, int device, int model_id, int N, const double *w,
                     const double *zn, const double *zn_err, int ndim, const double *lo,
                     const double *hi, const bisip_model_desc *desc);

/* Batch of independent spectra (BASELINE config 5): n_spectra spectra that share the model
 * shape, N and the prior box; w (E,N), zn (E,2,N), zn_err (E,2,N).  The same entry points
 * then take theta (E*Wp, ndim) with rows [e*Wp, (e+1)*Wp) belonging to spectrum e (W must
 * be a multiple of E).  PolynomialDecomposition: desc->taus/log_taus are shared by all
 * spectra.  One reference Inversion object per spectrum: src/bisip/models.py:41-57. */
int bisip_batch_create(bisip_ctx **out, int device, int model_id, int n_spectra, int N,
                       const double *w, const double *zn, const double *zn_err, int ndim,
                       const double *lo, const double *hi, const bisip_model_desc *desc);
int bisip_ctx_nspectra(const bisip_ctx *ctx);

void bisip_ctx_destroy(bisip_ctx *ctx);

/* Replace the prior box (strict inequalities, lo < theta < hi).  PolynomialDecomposition: the reduced form's
 * expansion points are chosen again for the new box; where their operands live in device memory (a batch; a lone
 * spectrum's compensated tier from degree 6 on) the call waits for the device's earlier work before it rewrites them. */
int bisip_ctx_set_bounds(bisip_ctx *ctx, const double *lo, const double *hi);

/* Choose the kernel formulation (BISIP_VARIANT_*). */
int bisip_ctx_set_variant(bisip_ctx *ctx, int variant);
int bisip_ctx_get_variant(const bisip_ctx *ctx);

/* theta (W,ndim) host -> logp (W,) host.  Synchronous. */
int bisip_logprob(bisip_ctx *ctx, const double *theta, int64_t W, double *logp);

/* Same with device-resident buffers, asynchronous on `stream` (a hipStream_t;
 * NULL = the default stream). */
int bisip_logprob_dev(bisip_ctx *ctx, const double *d_theta, int64_t W, double *d_logp,
                      void *stream);

/* theta (W,ndim) -> Z (W,2,N). */
int bisip_forward(bisip_ctx *ctx, const double *theta, int64_t W, double *Z);
int bisip_forward_dev(bisip_ctx *ctx, const double *d_theta, int64_t W, double *d_Z,
                      void *stream);
/* Batch context: forward of rows that ALL belong to spectrum `spectrum` (the model response over one
 * spectrum's chain: utils.get_model_percentile per spectrum of a survey, src/bisip/utils.py:17-35). */
int bisip_forward_spectrum_dev(bisip_ctx *ctx, int64_t spectrum, const double *d_theta, int64_t W,
                               double *d_Z, void *stream);
/* The same responses written COLUMN-major, d_cols (n_spectra, 2N, W / n_spectra): one contiguous column per
 * (spectrum, part, frequency) -- the layout bisip_columns_percentiles_dev reads, so the model-space bands of
 * a survey need no transposition.  Any number of rows per spectrum. */
int bisip_forward_columns_dev(bisip_ctx *ctx, int64_t first_spectrum, int64_t n_spectra, const double *d_theta,
                              int64_t W, double *d_cols, void *stream);
/* ... or to n_spectra consecutive spectra starting with first_spectrum, W / n_spectra rows each
 * (a multiple of 64 when n_spectra > 1), in one launch. */
int bisip_forward_spectra_dev(bisip_ctx *ctx, int64_t first_spectrum, int64_t n_spectra, const double *d_theta,
                              int64_t W, double *d_Z, void *stream);

/* Gaussian log-likelihood of a model response the caller evaluated itself: Z (W,2,N)
 * [row 0 real, row 1 imaginary per walker] -> (W,)  =  -0.5*sum((zn - Z)^2/zn_err^2 +
 * 2 ln zn_err^2).  Inversion._log_likelihood(theta, f, x, y, yerr) with a callable f that is
 * not the model's own forward: src/bisip/models.py:59-62 (the host runs f, the reduction is
 * this kernel).  No prior.  Single-spectrum contexts only. */
int bisip_loglike_z(bisip_ctx *ctx, const double *Z, int64_t W, double *loglike);
int bisip_loglike_z_dev(bisip_ctx *ctx, const double *d_Z, int64_t W, double *d_loglike,
                        void *stream);

/* ---- device-resident stretch move (the emcee inner loop the reference runs around the
 * log-probability: src/bisip/models.py:111-118; algorithm: emcee StretchMove with a
 * red/blue split, a = stretch scale) -------------------------------------------------
 * One half-step moves the walkers of the active half.  Slot t moves walker active[t]
 * along the line through walker partner[t] (a member of the complementary half):
 *     q = c - (c - s)*zz[t];   accept iff factor[t] + logp(q) - logp(s) > logu[t]
 * where the caller supplies factor = (ndim-1)*ln(zz) and logu = ln(u) from its RNG
 * stream (bisip_amd/sampler.py documents the stream order).  All pointers are device
 * pointers; coords (W,ndim) and logp (W,) are updated in place; optional chain_row
 * (W,ndim) / logp_row (W,) receive the post-move state of the moved walkers and
 * naccept (W,) counts acceptances; status gets bit 0 set if a proposal's
 * log-probability is NaN. */
typedef struct bisip_stretch_args {
    double *coords;
    double *logp;
    const int32_t *active;
    const int32_t *partner;
    const double *zz;
    const double *factor;
    const double *logu;
    int64_t n_slots;   /* slots in this half-step                                    */
    int64_t slot_lo;   /* eval: first slot evaluated by this rank                    */
    int64_t slot_hi;   /* eval: one past the last slot evaluated by this rank        */
    double *block;     /* eval: out (slot_hi-slot_lo, ndim+2) = row, logp, accepted  */
                       /* apply: in, the all-gathered (world*pad, ndim+2) buffer     */
    double *chain_row; /* may be NULL */
    double *logp_row;  /* may be NULL */
    int32_t *naccept;  /* may be NULL */
    int32_t *status;   /* required */
    int64_t pad;       /* apply: rows per rank slab in `block`                       */
    int32_t world;     /* apply: number of ranks that produced `block`               */
    int64_t walkers_per_spectrum; /* batch contexts: Wp (even); walker i belongs to spectrum
                                     i / Wp.  Ignored (may be 0) for single-spectrum contexts */
} bisip_stretch_args;

/* Single-rank half-step: evaluate all n_slots slots and update the state (one launch). */
int bisip_stretch_half_dev(bisip_ctx *ctx, const bisip_stretch_args *args, void *stream);
/* Sharded half-step: evaluate slots [slot_lo, slot_hi) into args->block ... (single-spectrum contexts: a batch of
 * spectra shards as whole replicas; BISIP_EUNSUPPORTED for a batch context) */
int bisip_stretch_eval_dev(bisip_ctx *ctx, const bisip_stretch_args *args, void *stream);
/* ... and, after the caller's all-gather of the blocks, apply all n_slots slots. */
int bisip_stretch_apply_dev(bisip_ctx *ctx, const bisip_stretch_args *args, void *stream);

/* Run n_steps whole iterations (2 half-steps each) back to back on `stream` with no host
 * round trip.  `first` holds the pointers of step 0 / half 0; the random-stream arrays are
 * laid out (n_steps, 2, nh) with nh = (W+1)/2 (half 0 has ceil(W/2) slots, half 1
 * floor(W/2)); chain_row / logp_row advance by W*ndim / W per step.  For a batch context W
 * is the TOTAL number of walkers E*Wp and the arrays are (n_steps, 2, E, Wp/2).
 * thin_by >= 1: only every thin_by-th iteration is stored (chain_row / logp_row then hold
 * n_steps/thin_by rows; n_steps must be a multiple of thin_by). */
int bisip_stretch_run_dev(bisip_ctx *ctx, const bisip_stretch_args *first, int64_t W,
                          int64_t n_steps, int64_t thin_by, void *stream);

/* ---- the same move with the walkers sharded over the GPUs of one node (one process per GPU,
 * RCCL over xGMI; the reference's only parallel hook is fit(pool=...) -> emcee's pool.map over
 * walkers, src/bisip/models.py:84,91-94,115).  Every rank holds the whole ensemble and the same
 * random stream, evaluates slots [rank's block) of each half-step, and one all-gather per
 * half-step rebuilds the state on every rank.
 *
 * bisip_stretch_run_sharded_dev: n_steps whole iterations, per half-step
 *     eval (this rank's slots -> its slab of the gather buffer)  ->  ncclAllGather (in place)
 *     ->  apply (every slot -> state + chain)
 * all enqueued on `stream` with no host round trip.  Arguments as for bisip_stretch_run_dev
 * (first->block / pad / world / slot_lo / slot_hi are ignored: the library owns the gather
 * buffer, allocated once per context, ceil(slots/world)*(ndim+2) doubles per rank).  `comm` is
 * an ncclComm_t whose ranks all make this call with the same arguments: one made by
 * bisip_rccl_comm_create, or an existing one (e.g. PyTorch's ProcessGroupNCCL._comm_ptr()).
 * The chain is bit-identical to bisip_stretch_run_dev's on every rank.  RCCL is bound at run
 * time (dlopen of librccl.so.1); BISIP_EUNSUPPORTED when it is absent, BISIP_ERCCL when a call
 * fails.  Single-spectrum contexts only (a batch of spectra shards as whole replicas). */
int bisip_stretch_run_sharded_dev(bisip_ctx *ctx, void *comm, const bisip_stretch_args *first,
                                  int64_t W, int64_t n_steps, int64_t thin_by, void *stream);

/* Test aid: the same loop with EVERY rank of a `world`-rank group evaluated on this device, one after
 * another, each into the slab of the gather buffer it would have sent, and no collective -- the slot
 * ranges, pads and slab offsets a multi-rank run depends on, checkable on one GPU (odd ensembles, more ranks
 * than slots, empty shards).  The chain equals the fused single-GPU chain bit for bit. */
int bisip_stretch_run_sharded_sim_dev(bisip_ctx *ctx, int world, const bisip_stretch_args *first, int64_t W,
                                      int64_t n_steps, int64_t thin_by, void *stream);

/* Stand-alone communicator for callers without one: rank 0 fills a BISIP_RCCL_ID_BYTES id
 * (ncclGetUniqueId), ships it to the other ranks by any means, and every rank creates its
 * communicator on `device` (ncclCommInitRank: collective over the `world` ranks). */
#define BISIP_RCCL_ID_BYTES 128
int bisip_rccl_unique_id(void *id);
int bisip_rccl_comm_create(void **comm, int world, int rank, const void *id, int device);
int bisip_rccl_comm_destroy(void *comm);

/* A batch context that holds spectra [first_spectrum, first_spectrum + E) of a larger survey
 * (one block per GPU): bisip_stretch_draw_dev then keys spectrum e's stream by its index in the
 * SURVEY, so every spectrum's chain is the same however the survey is split over GPUs.
 * Default 0.  (The reference has no batch object: a survey is a loop over Inversion objects,
 * src/bisip/models.py:41-57.) */
int bisip_ctx_set_spectrum_offset(bisip_ctx *ctx, int64_t first_spectrum);

/* Fill the random-stream arrays on the device (counter-based Philox4x32-10; the contract
 * is documented in bisip_amd/csrc/sampler_kernels.h and bisip_amd/sampler.py).
 * d_perm (n_steps,3) = per-step affine split (A, A^-1 mod W, B).  W = walkers per ensemble;
 * a batch context draws for all its spectra: arrays (n_steps, 2, E, W/2), global walker ids. */
int bisip_stretch_draw_dev(bisip_ctx *ctx, int64_t W, double a, uint64_t seed, int64_t step0,
                           int64_t n_steps, const int32_t *d_perm, int32_t *d_active,
                           int32_t *d_partner, double *d_zz, double *d_factor, double *d_logu,
                           void *stream);

/* bisip_stretch_run_dev for the Philox stream WITHOUT its arrays: every half-step launch draws its slots' entries
 * (walker, partner, z, (ndim-1) ln z, ln u) from their counters in place -- the same numbers bisip_stretch_draw_dev
 * writes, so the chain is that of draw + run bit for bit -- and nothing of the stream is written, stored or read:
 * at a million walkers the arrays are 34 MB per iteration, a fifth of the half-step's traffic, and drawing them
 * took a quarter of its time.  first->active / partner / zz / factor / logu are ignored (may be null); a, seed,
 * step0, d_perm as for bisip_stretch_draw_dev (d_perm: this chunk's n_steps rows).  Only where
 * bisip_stretch_philox_inline(ctx, W) returns 1 -- a single ensemble big enough for the packed-state half-step
 * (ndim <= 7, >= 131,072 walkers) -- otherwise BISIP_EUNSUPPORTED: draw, then run.  (emcee draws its stream on the
 * host, src/bisip/models.py:111-118 -> EnsembleSampler.run_mcmc.) */
int bisip_stretch_philox_inline(const bisip_ctx *ctx, int64_t W);
int bisip_stretch_run_philox_dev(bisip_ctx *ctx, const bisip_stretch_args *first, int64_t W, int64_t n_steps,
                                 int64_t thin_by, double a, uint64_t seed, int64_t step0, const int32_t *d_perm,
                                 void *stream);

/* Persistent sampler: one workgroup per ensemble runs n_steps iterations inside ONE launch
 * (ensemble in LDS, workgroup barrier between half-steps, several lanes per walker for the
 * models with a frequency loop).  Reads the same pre-drawn random stream as
 * bisip_stretch_run_dev -- arrays (n_steps, 2, E, nh), nh = (walkers_per_ensemble+1)/2,
 * global walker ids -- and produces bit-identical results.  Requires
 * walkers_per_ensemble*(ndim+1)*8 <= 65536 bytes of LDS and ceil(walkers_per_ensemble/2)
 * <= 512; otherwise returns BISIP_EUNSUPPORTED (use bisip_stretch_run_dev).
 * n_walkers = n_ensembles * walkers_per_ensemble. */
typedef struct bisip_persist_args {
    double *coords;              /* (n_walkers, ndim) in/out */
    double *logp;                /* (n_walkers,)      in/out */
    int64_t n_walkers;
    int64_t walkers_per_ensemble;
    int64_t n_steps;
    int64_t thin_by;
    const int32_t *active;       /* random stream, as in bisip_stretch_args */
    const int32_t *partner;
    const double *zz;
    const double *factor;
    const double *logu;
    double *chain;               /* (n_steps/thin_by, n_walkers, ndim) or NULL */
    double *logp_chain;          /* (n_steps/thin_by, n_walkers) or NULL */
    int32_t *naccept;            /* (n_walkers,) or NULL */
    int32_t *status;             /* required */
} bisip_persist_args;
int bisip_stretch_persistent_dev(bisip_ctx *ctx, const bisip_persist_args *args, void *stream);

/* Posterior mean and (population) standard deviation of every parameter, per ensemble, of a
 * chain resident in device memory -- the device form of get_param_mean / get_param_std
 * (src/bisip/utils.py:55-85: np.mean / np.std over the flattened chain) for batches whose
 * chains are too big to be worth copying to the host.
 * d_chain points at the first sample to use; sample k is at d_chain + k*sample_stride
 * doubles and holds (n_ensembles*walkers_per_ensemble, ndim) rows, ensemble e owning rows
 * [e*Wp, (e+1)*Wp).  (discard/thin of get_chain: offset the pointer, multiply the stride.)
 * d_mean, d_std: (n_ensembles, ndim).  d_work: bisip_chain_moments_workspace() doubles.
 * Two passes (mean, then centred squares), fixed summation order, asynchronous on stream. */
int64_t bisip_chain_moments_workspace(int64_t n_samples, int64_t n_ensembles, int ndim);
int bisip_chain_moments_dev(const double *d_chain, int64_t n_samples, int64_t sample_stride,
                            int64_t n_ensembles, int64_t walkers_per_ensemble, int ndim,
                            double *d_mean, double *d_std, double *d_work, void *stream);

/* Percentiles of every parameter, per ensemble, of a chain resident in device memory -- the
 * device form of get_param_percentile (src/bisip/utils.py:37-53: np.percentile(chain, p,
 * axis=0), linear interpolation; the reference's default p is [2.5, 50, 97.5]).  Chain layout,
 * d_chain / sample_stride conventions as for bisip_chain_moments_dev.  percentiles: host array
 * (n_percentiles,) in [0, 100].  d_out: (n_percentiles, n_ensembles, ndim).  d_work:
 * bisip_chain_percentiles_workspace() BYTES of device memory (two column-major copies of the
 * used samples + the sort's scratch; 0 is returned for a shape that is not supported:
 * more than 2^31 values).  At most 8 percentiles per call are found by selecting the order statistics they
 * need (asynchronous on stream), more by sorting the columns (after a short synchronous upload); the doubles
 * are numpy.percentile's either way. */
int64_t bisip_chain_percentiles_workspace(int64_t n_samples, int64_t n_ensembles,
                                          int64_t walkers_per_ensemble, int ndim, int n_percentiles);
int bisip_chain_percentiles_dev(const double *d_chain, int64_t n_samples, int64_t sample_stride,
                                int64_t n_ensembles, int64_t walkers_per_ensemble, int ndim,
                                const double *percentiles, int n_percentiles, double *d_out,
                                void *d_work, int64_t work_bytes, void *stream);

/* np.percentile(rows, p, axis=0) for a device-resident (n_rows, n_cols) array (linear rule):
 * d_out (n_percentiles, n_cols).  Workspace in BYTES (0: more than 2^31 values). */
int64_t bisip_column_percentiles_workspace(int64_t n_rows, int n_cols, int n_percentiles);
int bisip_column_percentiles_dev(const double *d_rows, int64_t n_rows, int n_cols,
                                 const double *percentiles, int n_percentiles, double *d_out,
                                 void *d_work, int64_t work_bytes, void *stream);

/* np.percentile of n_columns contiguous columns of n values each (d_cols (n_columns, n): what
 * bisip_forward_columns_dev writes): d_out (n_percentiles, n_columns).  Selection of the order statistics,
 * no workspace, asynchronous on stream. */
int bisip_columns_percentiles_dev(const double *d_cols, int64_t n_columns, int64_t n, const double *percentiles,
                                  int n_percentiles, double *d_out, void *stream);

/* The same for n_groups stacked arrays (n_groups, n_rows, n_cols) in one sort: d_out
 * (n_percentiles, n_groups, n_cols) -- the model responses of many spectra's chains at once. */
int64_t bisip_grouped_percentiles_workspace(int64_t n_groups, int64_t n_rows, int n_cols, int n_percentiles);
int bisip_grouped_percentiles_dev(const double *d_rows, int64_t n_groups, int64_t n_rows, int n_cols,
                                  const double *percentiles, int n_percentiles, double *d_out,
                                  void *d_work, int64_t work_bytes, void *stream);

/* Percentiles of the MODEL response over a chain -- utils.get_model_percentile
 * (src/bisip/utils.py:17-35: a Python loop of forward() over the chain, then np.percentile
 * over axis 0) in one call: theta (W, ndim) host -> forward on the device -> per-(part,
 * frequency) percentiles on the device -> out (n_percentiles, 2, N) host.  Only theta goes up
 * and n_percentiles*2N doubles come back (the responses are written column by column and the order
 * statistics selected from the columns: the same doubles as np.percentile).  Single-spectrum contexts. */
int bisip_forward_percentiles(bisip_ctx *ctx, const double *theta, int64_t W,
                              const double *percentiles, int n_percentiles, double *out);

/* Host: the stretch move's random stream in numpy.random.RandomState order for n_steps
 * iterations of a W-walker ensemble (the contract is bisip_amd/sampler.py:draw_step).
 * mt_key[624] / *mt_pos are RandomState.get_state()[1:3], advanced in place exactly as
 * NumPy would.  Outputs are (n_steps, 2, (W+1)/2): active and partner walker ids, stretch
 * factor zz and accept uniform u (the caller takes the logs). */
int bisip_numpy_stretch_stream(uint32_t *mt_key, int32_t *mt_pos, int64_t W, double a,
                               int64_t n_steps, int32_t *active, int32_t *partner, double *zz,
                               double *u);

/* Host, after the fact: how far are log-probabilities that a QR-reduced kernel produced from the
 * reduced form evaluated from the unrounded operands (in long double; in binary128 for a spectrum on the
 * compensated tier, whose operands come from a QR in binary128)?  theta (W, ndim) and logp (W,)
 * are host arrays -- typically a sampler's final ensemble and its log-probabilities (batch context:
 * the (E*Wp, ndim) layout of bisip_logprob).  *worst_rel = max |logp - reference| / max(1, |reference|)
 * over the rows inside the prior.  BISIP_VARIANT_AUTO picks a formulation from an ESTIMATE made on
 * probe rows when the context is created; this measures the same quantity where the walkers ended
 * up.  PolynomialDecomposition contexts only (BISIP_EUNSUPPORTED otherwise). */
int bisip_ctx_reduced_check(bisip_ctx *ctx, const double *theta, int64_t W, const double *logp, double *worst_rel);

/* bisip_logprob (the host-buffer entry emcee calls) measures the QR-reduced kernel it ran on up to 256
 * rows of the caller's own batch -- on a context's first call and every 2^n-th after it -- the way
 * bisip_ctx_reduced_check does.  Past 2e-11 a context on BISIP_VARIANT_AUTO moves to the next
 * formulation (compensated, then per-frequency) and evaluates the batch again with it; the choice holds
 * until bisip_ctx_set_bounds.  A caller-forced variant is measured and left alone.  enable: 1 / 0 turn
 * the guard on (default) / off, anything else leaves it; outputs (each may be NULL): checks made so far,
 * the worst relative error any of them saw, how many times the context changed formulation.
 * The device-pointer entry bisip_logprob_dev never synchronises and cannot guard itself: its callers hold
 * the rows and bring some of them to bisip_ctx_reduced_guard_rows (the device sampler does, chunk by chunk,
 * before it keeps a chunk).
 * The guard's findings are CONTEXT STATE: an escalation is remembered by the context, and every later launch on
 * it -- bisip_logprob_dev and the stretch-move entries included -- runs the formulation the guard moved to.  A
 * context that only ever sees device-pointer calls keeps the formulation its estimate chose; so a host-loop
 * sampler (guarded) and a device sampler on two contexts of the same spectrum may, after an escalation, run
 * different kernels (both within the tolerance the guard enforces; the chains then differ in the last bits).
 * bisip_logprob updates that state and the context's workspace: like every entry that takes a context it is
 * not re-entrant for ONE context (contexts are not shared across threads without locking). */
int bisip_ctx_reduced_guard(bisip_ctx *ctx, int enable, int64_t *n_checks, double *worst_rel, int *escalations);

/* The guard for callers whose rows live on the DEVICE (bisip_logprob_dev and the stretch-move entries never
 * synchronise, so they cannot measure themselves): the caller brings a few rows and the log-probabilities the
 * context's kernel gave them to the host -- the device sampler brings rows of its initial ensemble and, chunk by
 * chunk, the stored samples nearest to the shell logp = 0 (bisip_chain_shell_rows_dev), and does so BEFORE it
 * keeps a chunk -- and this call measures them as bisip_ctx_reduced_check does (rows outside the prior and NaN
 * rows are skipped; batch context: the (E*m, ndim) layout, m rows per spectrum).  *worst_rel: the measurement.
 * *escalated = 1: it was past 2e-11, the context runs on BISIP_VARIANT_AUTO and has just moved to the next
 * formulation, exactly as bisip_logprob's guard moves it -- the caller's log-probabilities of this and every
 * later row are stale: re-evaluate its state and re-run what it ran (the device sampler restores the chunk's
 * initial state and runs the chunk again: its random stream is counter-based or saved).  A caller-forced
 * variant, a context whose guard is off, or one that already runs the per-frequency form: measured (or not
 * at all), never moved.  PolynomialDecomposition contexts only (BISIP_EUNSUPPORTED otherwise). */
int bisip_ctx_reduced_guard_rows(bisip_ctx *ctx, const double *theta, int64_t W, const double *logp,
                                 double *worst_rel, int *escalated);

/* Per ensemble the k stored samples whose |log-probability| is smallest -- where the relative parity tolerance
 * max(1, |logp|) has denominator 1 and a kernel's absolute error shows -- of a chain resident in device memory:
 * d_chain (n_samples, n_ensembles*walkers_per_ensemble, ndim), d_logp (n_samples, n_ensembles*walkers_per_ensemble),
 * both contiguous.  d_out (n_ensembles, k + n_stride, ndim + 1): each row = a sample's theta followed by its
 * log-probability; slots no sample fills (fewer than k finite log-probabilities) are NaN rows.  Selection on
 * the bits of |logp| to 24 bits (exponent + 13 bits of mantissa): every sample nearer than the k-th is
 * there; of the samples that tie with the k-th to that resolution (1e-4 relative) the first to arrive fill the
 * remaining slots (ties != 0) or none does (ties == 0: the selected SET is then the same in every run and on
 * every rank of a sharded run, which therefore take the same decision from it with no collective).
 * n_stride > 0 (<= 256, <= walkers_per_ensemble) appends n_stride evenly spaced walkers of the FIRST sample.  d_work:
 * bisip_chain_shell_rows_workspace(n_ensembles) bytes.  Asynchronous on stream.  (No reference counterpart:
 * it serves the guard above; the run it guards is src/bisip/models.py:111-118.) */
int64_t bisip_chain_shell_rows_workspace(int64_t n_ensembles);
int bisip_chain_shell_rows_dev(const double *d_chain, const double *d_logp, int64_t n_samples, int64_t n_ensembles,
                               int64_t walkers_per_ensemble, int ndim, int k, int n_stride, int ties, double *d_out,
                               void *d_work, void *stream);

/* Sums and second moments of an ensemble's positions, shifted by walker 0: d_coords (W, ndim) device ->
 * d_out (ndim + ndim (ndim + 1) / 2) device = S_j = sum_i (x_ij - x_0j), then P_jk = sum_i (x_ij - x_0j)(x_ik - x_0k)
 * for k >= j, row by row.  What emcee's initial-state test -- the condition number of the centred, column-scaled
 * positions, raised inside emcee.EnsembleSampler.run_mcmc (src/bisip/models.py:111-118) -- needs of a big
 * ensemble that is on the device anyway: the host forms the centred Gram matrix P - S S^T / W and decides
 * (bisip_amd/sampler.py:walkers_independent); a NaN or an inf anywhere makes the sums non-finite.  ndim <= 8
 * (BISIP_EUNSUPPORTED beyond: the sums live in registers).  d_work: bisip_ensemble_gram_workspace() doubles.
 * Fixed summation order; asynchronous on stream. */
int64_t bisip_ensemble_gram_workspace(int64_t W, int ndim);
int bisip_ensemble_gram_dev(const double *d_coords, int64_t W, int ndim, double *d_out, double *d_work, void *stream);

/* Host: read n_files 5-column spectrum files (freq, amp, pha, amp_err, pha_err; comma separated,
 * `headers` lines skipped, '#' comments and blank lines ignored -- what the reference reads one
 * at a time with np.loadtxt(skiprows=headers, delimiter=','), src/bisip/utils.py:121-123) on
 * `threads` host threads (<= 0: as many as the process may use, at most 16) into tables
 * (n_files, n_rows, 5).  status[i] = 0: table i is filled
 * with exactly the doubles np.loadtxt yields; 1: file i is not in the plain format, cannot be
 * read, or does not hold n_rows rows -- read that one the reference's way (the Python host does,
 * so such files behave and fail as in the reference).  No GPU involved. */
int bisip_read_tables(const char *const *paths, int64_t n_files, int headers, int64_t n_rows,
                      double *tables, int32_t *status, int threads);

/* Host: one Philox4x32-10 block (counter[4], key[2]) -> out[4]; for known-answer tests. */
void bisip_philox4x32(const uint32_t *counter, const uint32_t *key, uint32_t *out);

/* Introspection */
int bisip_ctx_ndim(const bisip_ctx *ctx);
int bisip_ctx_nfreq(const bisip_ctx *ctx);
int bisip_ctx_device(const bisip_ctx *ctx);
/* Which loop the per-frequency models (ColeCole, Shin, Dias2000) run for the current prior box and frequencies; bits:
 *   1  shared reciprocals, exponents unclamped: the box keeps every denominator product normal (ColeCole with
 *      one or two modes, Shin and Dias2000: one reciprocal per pair of frequencies 2k, 2k+1 -- up to four
 *      denominators -- and a last unpaired frequency its own; ColeCole with three modes and more: one per group
 *      of up to four of a frequency's denominators);
 *   2  (only with 1; ColeCole up to three modes, Shin) geometric frequency grid: SOME spectrum's
 *      ln w_{16k+q} = ln w_{16k} + q*step (q < 16) to 4e-15 (bisip_frequency_grid_step), and every spectrum that is on
 *      such a grid takes its exponentials once per block of sixteen frequencies and steps them by multiplication
 *      (a batch decides per spectrum).  The environment variable BISIP_NO_GRID, read when a context is
 *      created, switches bit 2 off (measurement aid).
 * 0 for PolynomialDecomposition and for boxes widened past the limits. */
int bisip_ctx_loop_flags(const bisip_ctx *ctx);
/* Host-only: 1 and *step = the common step of ln w when w[0..N) is such a grid (N >= 8), else 0 and *step = 0. */
int bisip_frequency_grid_step(int N, const double *w, double *step);
/* Walker-independent part of the log-likelihood, -0.5*sum(2*ln(sigma^2)). */
double bisip_ctx_loglike_const(const bisip_ctx *ctx);
/* Name of the kernel bisip_logprob_dev launches for the current variant. */
const char *bisip_ctx_kernel_name(const bisip_ctx *ctx);
/* PolynomialDecomposition: worst relative log-probability error of the QR-reduced kernel for the
 * current prior box, estimated by emulating its double arithmetic on the host against long
 * double on ~200 probe rows per spectrum -- uniform in the box, clouds of small coefficients,
 * clouds around the least-squares solution and draws from the Gaussian posterior, the flat
 * valley where a sampler's walkers sit -- (0 for other models).  BISIP_VARIANT_AUTO runs the plain
 * reduced kernel while its estimate is <= 1e-12 (and 2N >= poly_deg+2), else the compensated one
 * while ITS estimate is <= 1e-12, else the collapsed form.  The value returned is the estimate of
 * the reduced kernel the current variant runs (or, when that is the collapsed form, of the better
 * of the two). */
double bisip_ctx_reduced_error(const bisip_ctx *ctx);

/* PolynomialDecomposition: how many spectra of the context run the plain (*n_plain) and the compensated
 * (*n_comp) QR-reduced kernel at the moment; both 0 when the per-frequency form runs.  A batch on
 * BISIP_VARIANT_AUTO decides per spectrum -- every spectrum runs what a context of its own would run, inside
 * one launch; a forced variant, and a batch whose mix bisip_logprob's guard closed, run one tier for all. */
int bisip_ctx_reduced_tiers(const bisip_ctx *ctx, int64_t *n_plain, int64_t *n_comp);

/* Host-only inspection of the walker-independent PolynomialDecomposition operands the
 * context precomputes (no GPU needed; used by the CPU-side tests).  Outputs:
 * G_re/G_im (N, P+1);  R (n,n) upper triangle row-major, bhat (n,), e (n,), rest (1,)
 * with n = P+2 (see BISIP_VARIANT_REDUCED);  lconst (1,) = -0.5*sum(2 ln sigma^2). */
int bisip_polydecomp_operands(int N, const double *w, const double *zn, const double *zn_err,
                              const bisip_model_desc *desc, double *G_re, double *G_im,
                              double *R, double *bhat, double *e, double *rest,
                              double *lconst);

/* Host-only: the error estimates behind BISIP_VARIANT_AUTO for one spectrum and prior box (what
 * bisip_ctx_reduced_error reports for a context): worst relative log-probability error of the
 * plain (est[0]) and the compensated (est[1]) QR-reduced kernel, from emulating their double
 * arithmetic against long double on the probe rows.  No GPU needed. */
int bisip_polydecomp_reduced_estimates(int N, const double *w, const double *zn, const double *zn_err,
                                       const bisip_model_desc *desc, const double *lo,
                                       const double *hi, double *est);

/* Measurement aid (no reference counterpart): ONE wavefront that reads the shader clock counter
 * (s_memtime) and the constant 100 MHz counter (s_memrealtime), idles for window_us and reads both
 * again; d_out (4 x int64, device) = shader ticks begin/end, 100 MHz ticks begin/end.  Enqueued on a
 * stream of its own beside a kernel under measurement it tells the engine clock the chip holds under
 * that kernel's load -- fp64-dense kernels run at 1.9-2.1 GHz, not at the 2.4 GHz the issue peak is
 * quoted at (benchmarks/micro/collapsed_r3.hip) -- without touching the kernel itself. */
int bisip_clock_probe_dev(int64_t *d_out, double window_us, void *stream);

/* Measurement aid (no reference counterpart): the rate a stream of INDEPENDENT fp64 FMAs reaches on this chip --
 * no dependency to wait for, no memory, 8 waves per SIMD on every compute unit, operands with full mantissas --
 * i.e. the ceiling a compute-bound log-probability kernel can be held to: 0.82-0.89 of the nominal
 * 1024 SIMDs x 2.4 GHz / 4 cycles with real data (the chip holds 2.2-2.3 GHz under it and a wave-instruction
 * takes 4.25-4.45 cycles; benchmarks/micro/fp64_stream_ceiling.hip).  One launch: every wave issues 32 * rounds
 * v_fma_f64; *wave_instructions (host, optional) = the launch's total.  d_out: bisip_fp64_stream_probe_lanes()
 * doubles on the device (written so that nothing is optimised away).  The caller times the launch. */
int64_t bisip_fp64_stream_probe_lanes(void);
int bisip_fp64_stream_probe_dev(double *d_out, int rounds, int64_t *wave_instructions, void *stream);

/* Host-only yardstick (no GPU, no prior): the PolynomialDecomposition log-likelihood
 * -0.5 (chi^2 + sum 2 ln sigma^2) of src/bisip/models.py:59-62 + cython_funcs.pyx:75-94 for W rows of theta,
 * from the QR-reduced form with NOTHING rounded to double on the way: kernel sums, Householder QR and
 * Q^T y in x87 long double, every row q_i - sum_j R_ij b_j accumulated as an unevaluated sum of two long
 * doubles.  Agrees with a 60-digit evaluation of the reference's per-frequency formula to ~1e-12 where
 * the reference's own double arithmetic is 1e-9 ... 1e-7 away (rows on the shell logp = 0 of degree 8-10
 * designs; tests/test_oracle_golden.py pins it with mpmath).  It is what bisip_ctx_reduced_check,
 * bisip_logprob's guard and the estimate behind BISIP_VARIANT_AUTO measure against.  Needs 2N >= poly_deg+2. */
int bisip_polydecomp_reduced_reference(int N, const double *w, const double *zn, const double *zn_err,
                                       const bisip_model_desc *desc, const double *theta, int64_t W, double *logp);

int bisip_abi_version(void);
int bisip_device_count(void);
const char *bisip_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* BISIP_HIP_H */
