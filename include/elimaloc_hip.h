/*
 * elimaloc_hip.h -- C ABI of the MI355X-native pcm_matching registration hot path.
 *
 * This is the drop-in boundary: every entry point replaces one in-process C++ call of the reference
 * (ELiMaLoc @ 2025-02-27).  Citations are relative to /root/reference/src/app/localization/ :
 *   reg.hpp/reg.cpp = pcm_matching/{include,src}/registration.{hpp,cpp}
 *   vhm.hpp/vhm.cpp = pcm_matching/{include,src}/voxel_hash_map.{hpp,cpp}
 *   pcm.hpp/pcm.cpp = pcm_matching/{include,src}/pcm_matching.{hpp,cpp}
 *
 * Conventions
 *   - plain pointers and sizes only; 4x4 / 6x6 / 3x3 matrices are column-major doubles (Eigen's default).
 *   - every function returns an int status: 0 = ok, <0 = error (see elm_strerror); the status is SEPARATE
 *     from the algorithmic is_success flag of RunRegister.
 *   - a context owns one GPU (one process per GPU), one HIP stream and all scratch memory.  Calls on one
 *     context must be serialised by the caller (the reference serialises them with mutex_pcl_, pcm.cpp:199,357).
 *   - no CPU fallback exists: without a gfx950 device elm_ctx_create fails.
 */
#ifndef ELIMALOC_HIP_H
#define ELIMALOC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ELM_OK 0
#define ELM_ERR_INVALID -1     /* bad argument */
#define ELM_ERR_DEVICE -2      /* HIP runtime error (elm_last_error has the text) */
#define ELM_ERR_NO_DEVICE -3   /* no usable gfx950 device */
#define ELM_ERR_COMM -4        /* RCCL error / not initialised */
#define ELM_ERR_UNSUPPORTED -5 /* e.g. a host-fed stream with a communicator attached; a search index that cannot be built */
#define ELM_ERR_IO -6          /* file missing / unreadable */
#define ELM_ERR_ALLOC -7       /* host allocation failed */

#define ELM_MAX_ITER_TRACE 64

typedef struct elm_ctx elm_ctx;
typedef struct elm_map elm_map;   /* VoxelHashMap, device resident (vhm.hpp:89-335) */
typedef struct elm_scan elm_scan; /* one source scan (sensor frame), device resident */

/* IcpMethod (reg.hpp:60) */
enum { ELM_P2P = 0, ELM_GICP = 1, ELM_VGICP = 2, ELM_AVGICP = 3 };

/* POD mirror of RegistrationConfig (reg.hpp:62-85), same field names. */
typedef struct elm_reg_config {
    int32_t i_max_thread;        /* unused on the GPU; kept for API parity */
    int32_t icp_method;          /* ELM_P2P .. ELM_AVGICP */
    int32_t voxel_search_method; /* parsed but unused by the reference (pcm.cpp:175) */
    int32_t use_radar_cov;       /* reg.hpp:186-217: first iteration adds CalPointCov of the point under the initial guess, later ones I */
    int32_t max_iteration;
    int32_t b_debug_print;
    double gicp_cov_search_dist;
    double max_search_dist;
    double lm_lambda;
    double icp_termination_threshold_m;
    double min_overlap_ratio;
    double max_fitness_score;
    double doppler_trans_lambda;
    double range_variance_m;
    double azimuth_variance_deg;
    double elevation_variance_deg;
    double ego_to_lidar_trans[3];
    double ego_to_lidar_rot[9];
    double ego_to_imu_rot[9];
} elm_reg_config;

/* Fills the shipped defaults of config/localization.ini:80-105. */
void elm_reg_config_default(elm_reg_config* cfg);

/* Per-iteration record (optional; see elm_reg_result.trace). */
typedef struct elm_iter_trace {
    double JTJ[36]; /* column-major, before damping */
    double JTr[6];
    double residual_sum;
    double n_corr;
    double x[6];
    double step_norm;
    double T[16];
} elm_iter_trace;

/* elm_reg_result.path: the accumulate kernels of the call.  GRID (dense / two-level cell grid: P2P, GICP) and VOXEL_LISTS (VGICP, AVGICP)
 * are the fast kernels; LISTS / WALK are the fall-back index forms of maps no grid can hold (or ELM_KERNEL); PAIRS = the per-pair kernels
 * (use_radar_cov, ELM_CHECK=strict_pairs, or -- with one warning per map on stderr -- asymmetric covariances where the search is the plain
 * walk: 12-27 times slower); | SIDE_RECORDS: the map holds asymmetric covariances and the fast kernels carried their antisymmetric sums. */
#define ELM_PATH_GRID 1
#define ELM_PATH_LISTS 2
#define ELM_PATH_VOXEL_LISTS 3
#define ELM_PATH_WALK 4
#define ELM_PATH_PAIRS 5
#define ELM_PATH_SIDE_RECORDS 16

/* Outputs of one RunRegister (reg.cpp:274-418). */
typedef struct elm_reg_result {
    double T[16];         /* returned pose (column-major) */
    double fitness_score; /* written only on success, as the reference's out-param (reg.cpp:415); else 0 */
    double d_fitness;     /* Registration::d_fitness_score_ at return */
    double local_cov[36]; /* I unless GICP (reg.cpp:280,142) */
    int32_t is_success;
    int32_t iterations; /* executed iterations */
    int32_t gate;       /* 0 none, 1 empty map, 2 overlap ratio (reg.cpp:352), 3 fitness (reg.cpp:405) */
    int32_t path;       /* which kernels ran (ELM_PATH_*; 0: nothing iterated) */
    double n_corr_last; /* correspondences of the last executed iteration */
    double point_iterations; /* scan points processed x iterations */
    /* work counters summed over the executed iterations (for the algorithmic-bytes model): 0 unless
     * elm_ctx_set_work_counters(ctx, 1) */
    double n_cand_total;     /* candidate map points (P2P/GICP) or voxel means (VGICP/AVGICP) distance-tested */
    double n_occ_total;      /* occupied neighbour voxels visited */
    double fallback_blocks;  /* workgroup launches that took the un-staged path */
    double n_tested_total;   /* candidates whose distance was actually evaluated (after exact cell pruning) */
} elm_reg_result;

typedef struct elm_map_info {
    uint64_t n_input_points;
    uint64_t n_points; /* retained by AddPoints' spacing rule */
    uint64_t n_voxels;
    uint64_t hash_capacity;
    double voxel_size;
    int32_t max_points_per_voxel;
    int32_t has_voxel_cov;
    int32_t has_point_cov;
    int32_t layout_flags; /* bit 0: GICP payload as 64-byte {mean, normal, k} records (a point covariance of the form I - 0.999 n n^T
                           * has its inverse rebuilt as I + k n n^T; a point outside that form is flagged and reads its stored inverse),
                           * bit 1: the same for the voxel covariances of VGICP / AVGICP (clear: ELM_CHECK=full_records, all stored inverses),
                           * bit 2: the P2P / GICP cell grid is the two-level (tiled) form (box too large / sparse for one dense table),
                           * bit 3: no point covariance is flagged: GICP runs the kernels without the stored-inverse fallback and gathers its
                           *        pair fused, A = w I + (w k) n n^T (clear with ELM_CHECK=pair_nine at map build: nine entries of w C^-1),
                           * bit 4: the same for the voxel covariances (VGICP's pair; AVGICP gathers sum w and sum (w k) n n^T per point),
                           * bit 5: the face sublists are written for AVGICP's fused walk,
                           * bit 6: ... and some voxel is flagged: the fused walk skips its pairs and a fix-up launch over the marked
                           *        workgroups adds them (ELM_CHECK=avg_inline at map build: the nine-entry walk with its in-line fallback instead),
                           * bit 7: some flagged POINT covariance has an asymmetric stored inverse (rank-deficient neighbourhood, U != V in its
                           *        SVD): GICP's kernels on this map also write the 15 antisymmetric side sums per workgroup and the solve restores
                           *        all 36 entries of J^T M J (LDLT on the lower triangle, as the reference); ELM_CHECK=strict_pairs runs
                           *        the reference's per-pair arithmetic instead (the in-product checker, 12-27 times slower),
                           * bit 8: the same for the voxel covariances (VGICP / AVGICP). */
    uint64_t device_bytes;
    uint64_t n_query_voxels; /* cell grid: voxels of the dense statistics box; neighbourhood lists: query voxels (0 until built) */
    uint64_t nbr_entries;    /* cell grid: == n_points (every map point once); neighbourhood lists: ~27 x n_points */
    uint64_t index_bytes;    /* device bytes of the search structures the accumulate kernels read (built on first use) */
    uint64_t index_part_bytes[4]; /* ... by structure: [0] cell grid (blocks + offsets: P2P / GICP), [1] voxel-mean lists + per-voxel records
                                   * (VGICP; AVGICP without a face table), [2] AVGICP's face sublists + their table, [3] neighbourhood lists (the
                                   * fall-back index of P2P / GICP) */
    uint64_t n_list_voxels;       /* query voxels of the voxel-mean lists (0 until built) */
} elm_map_info;

/* ---------------------------------------------------------------- run-time switches --------------- */
/* The library reads FIVE environment variables (+ ELM_DEVICES, read by the C++ shims in include/elimaloc/).  None is needed in
 * production: the defaults are what every number in DESIGN.md is measured with; the others select shipping code paths that other maps
 * reach by themselves (so that tests can force them onto small maps) or run the in-product checkers of the fast forms.
 *   ELM_KERNEL          grid (default: dense / two-level cell grid for P2P / GICP, voxel-mean lists for VGICP / AVGICP) | lists (the
 *                       fall-back index of maps no grid can hold: per-query-voxel neighbourhood lists) | direct (the plain walk -- 27 hash
 *                       probes, every bucket point, float64: the in-kernel reference of the parity tests).  Read at elm_ctx_create.
 *   ELM_GRID            comma-separated: dense | tiled (forbid / force the two-level form), max_cells=N (cell budget of the dense offset
 *                       table, default 1.5e9), max_block_bytes=N (block arrays beyond N bytes are addressed in 16-byte units, default 4 GB).
 *                       Read when a map's search index is built.
 *   ELM_CHECK           comma-separated in-product checkers: strict_pairs (covariance methods run the reference's per-pair arithmetic: all
 *                       36 entries of J^T M J, 3x3 products and an inverse per pair), full_records (pairs read the stored 3x3 inverses
 *                       instead of the compact {mean, normal, k} records), pair_nine / avg_nine (nine entries of w C^-1 per pair instead of
 *                       the fused gathers), avg_inline / avg_fixup (AVGICP on maps with flagged voxels: in-line fallback / fix-up launch,
 *                       whatever the map's share of flagged voxels), query_direct (elm_map_get_correspondences by the plain walk).
 *   ELM_SCAN_ORDER      none: elm_scan_upload keeps the caller's point order (default: Hilbert order over 2 m cells, on the device).
 *   ELM_GROUP_EXCHANGE  host | rccl: the exchange of a device group (default: RCCL when every rank has a device of its own).
 *   ELM_DEVICES         (shims) "0,1,2,3": the process-wide context of the C++ shims is a device group over these GPUs.
 * Measured-negative experiments of rounds 2-5 (fused reduction, half-set streams, graph replay, wave-level reduction, previous-winner
 * bound, patch table ...) are not in the library any more: profiles/r06_removed_*.patch re-adds each. */

/* ---------------------------------------------------------------- context ------------------------- */
int elm_ctx_create(int device_id, elm_ctx** out);
void elm_ctx_destroy(elm_ctx* ctx);
const char* elm_last_error(const elm_ctx* ctx); /* text of the last ELM_ERR_DEVICE / ELM_ERR_COMM */
const char* elm_strerror(int status);
int elm_ctx_synchronize(elm_ctx* ctx);
/* native hipStream_t of the context (for hipEvent timing by the caller) */
void* elm_ctx_stream(elm_ctx* ctx);

/* Optional per-kernel timing with hipEvents recorded on the context stream around every accumulate launch and
 * every solve(+exchange) step of elm_register_batch* / _stream.  Totals accumulate until reset. */
typedef struct elm_profile {
    uint64_t accumulate_launches;
    uint64_t solve_steps;
    double accumulate_ms; /* sum of the spans of the main accumulate kernel (k_accumulate_cell / _vnbr / _direct) */
    double solve_ms;      /* sum of the reduce/solve (+ all-reduce, + slot refill) spans */
} elm_profile;
int elm_ctx_set_profiling(elm_ctx* ctx, int enable);
/* Work counters of elm_reg_result (n_cand_total, n_occ_total, n_tested_total, fallback_blocks): OFF by default -- the registration
 * launches then carry no instrumentation and those fields read 0; on: the same kernels with
 * the counters compiled in (~1 % slower).  poses, flags, iteration counts and point_iterations do not depend on the switch. */
int elm_ctx_set_work_counters(elm_ctx* ctx, int enable);
int elm_ctx_get_profile(elm_ctx* ctx, elm_profile* out, int reset);

/* ---------------------------------------------------------------- map ----------------------------- */
/* VoxelHashMap::Init + AddPoints (vhm.cpp:26-29, 270-285; call site pcm.cpp:87-88).  xyz: n*3 float32 map
 * points in file order (the PCD is float32, pcm.hpp:205-215).  Serial insertion semantics (trunc keys,
 * min-spacing rule, <= max_points) are reproduced per voxel; buckets are uploaded to the device. */
int elm_map_build(elm_ctx* ctx, const float* xyz, size_t n, double voxel_size, int max_points_per_voxel,
                  elm_map** out);
void elm_map_destroy(elm_map* map);
/* VoxelHashMap::CalVoxelCovAll (vhm.hpp:183-193), HIP kernel over voxels */
int elm_map_cal_voxel_cov_all(elm_map* map);
/* VoxelHashMap::CalPointCovAll (vhm.hpp:252-257), HIP kernel over map points */
int elm_map_cal_point_cov_all(elm_map* map, double d_search_dist);
/* Build the search index of the P2P / GICP correspondence search now instead of on the first registration (an init-time cost of a few
 * seconds on a 10 M-point map): the dense half-voxel CELL GRID -- the map points once more, sorted by cell in 48-byte blocks of four,
 * plus one offset per cell of the bounding box -- when the box fits the cell and byte budgets; its two-level form (tiles of 8 x 8 columns
 * with their own z range) when it does not; the round-1 neighbourhood lists (per query voxel the points of its 27 buckets, 27x the map)
 * only when neither can be built or ELM_KERNEL=lists asks for them.  Results do not depend on which index is in use (the reference's
 * visiting order -- vhm.cpp:234-240, insertion order inside a bucket -- settles exact ties in all of them). */
int elm_map_build_neighbourhoods(elm_map* map);
int elm_map_get_info(const elm_map* map, elm_map_info* info);
/* VoxelHashMap::Empty (vhm.hpp:325) */
int elm_map_empty(const elm_map* map);
/* VoxelHashMap::Pointcloud (vhm.cpp:245-255): xyz[3n] doubles; optional per-point cov (9, col-major) and mean (3).
 * Order is bucket order (the reference's order is unordered_map iteration order; only the set is contractual). */
int elm_map_download_points(const elm_map* map, double* xyz, double* cov9, double* mean3, size_t cap);
/* all voxels: stored key (3 ints), point count, cov (9) and mean (3); Covariances() (vhm.cpp:257-265) is the
 * subset with count > 2 */
int elm_map_download_voxels(const elm_map* map, int32_t* key3, int32_t* npts, double* cov9, double* mean3,
                            size_t cap);
/* VoxelHashMap::GetCorrespondencePoints (what = 0; voxel_hash_map.cpp:31-88), ::GetCorrespondencesCov (1; :90-151) and
 * ::GetCorrespondencesAllCov (2; :153-206) as calls of their own -- Registration::RunRegister (registration.cpp:317-334) never needs them
 * (its iterations search and accumulate in one kernel), a caller of the map's public interface may.  xyz: n points in the MAP frame
 * (PointStruct::pose after TransformPoints), column order x, y, z.  Pairs come back in input order (the reference's vectors: ranges joined
 * in order): src_index[k] = the query point of pair k, tgt_index[k] = its target -- what 0: position of the map point in
 * elm_map_download_points order, what 1 / 2: position of the voxel in elm_map_download_voxels order; -1 = the reference's default target at
 * the origin (no neighbour bucket at all around the point, voxel_hash_map.cpp:37 / :105; never for what 2).  what 2 yields up to seven
 * pairs per point in the reference's neighbour order (0, +x, -x, +y, -y, +z, -z).  At most `cap` pairs are written; *n_pairs is the full
 * count.  The search is the accumulate kernels' own (their QUERY instantiations); ELM_CHECK=query_direct runs the plain 27-probe walk instead. */
int elm_map_get_correspondences(elm_ctx* ctx, const elm_map* map, int what, const double* xyz, size_t n, double max_dist,
                                uint32_t* src_index, int32_t* tgt_index, size_t cap, size_t* n_pairs);

/* Registration::AlignCloudsLocal (method ELM_P2P; registration.cpp:15-66), ::AlignCloudsLocalPointCov (ELM_GICP; :68-152) and
 * ::AlignCloudsLocalVoxelCov (ELM_VGICP / ELM_AVGICP; :154-225) on pairs the CALLER holds (RunRegister pairs and accumulates in one kernel and
 * never calls them): src_local = PointStruct::local of the n source points (sensor frame), tgt_xyz = the targets' positions in the map
 * frame -- PointStruct::pose for P2P, covariance.mean for the covariance methods (registration.cpp:97, 172) --, tgt_cov9 = their 3x3
 * covariances (column-major; NULL for P2P), src_cov9 = the source points' covariance terms, added when cfg->use_radar_cov (NULL: none).
 * T_out = the step as a column-major 4x4 (the reference's return value), local_cov = the inverse of the damped normal matrix (written
 * for ELM_GICP only, like the reference's out-parameter), *fitness_score = d_fitness_score_ (residual sum / n).  cfg: lm_lambda and
 * use_radar_cov are read (NULL: the defaults).  The pairs are accumulated on the device with the reference's per-pair arithmetic
 * (all 36 entries of J^T M J for the covariance methods: a non-symmetric covariance behaves as in the reference). */
int elm_align_clouds_local(elm_ctx* ctx, int method, const double* src_local, const double* tgt_xyz, const double* tgt_cov9,
                           const double* src_cov9, size_t n, const double last_icp_pose[16], double trans_th,
                           const elm_reg_config* cfg, double T_out[16], double local_cov[36], double* fitness_score);

/* VoxelHashMap::FindGroundHeight (vhm.hpp:285-322); *found = 0/1 */
int elm_map_find_ground_height(const elm_map* map, double x, double y, double* ground_z, int* found);

/* ---------------------------------------------------------------- scans --------------------------- */
/* Upload one source scan (sensor frame, packed float32 xyz, 12 bytes per point; PointStruct.local == .pose, pcm.hpp:205-220).
 * n_total is the size of the whole scan when this context holds only a shard of it (multi-GPU; the overlap
 * ratio of reg.cpp:351 is taken against n_total); pass n_total = n on one GPU.  The points go to HBM as they are and are
 * re-ordered ON THE DEVICE along a Hilbert curve over 2 m sensor-frame cells for locality (source order is not contractual:
 * the reference's own VoxelDownsample emits unordered_map order, vhm.hpp:278-280; ELM_SCAN_ORDER=none keeps the caller's). */
int elm_scan_upload(elm_ctx* ctx, const float* xyz, size_t n, size_t n_total, elm_scan** out);
void elm_scan_destroy(elm_scan* scan);
size_t elm_scan_size(const elm_scan* scan);
/* The resident points of a scan (xyz[3 * min(cap, size)] float32, in the device order): e.g. the undistorted, downsampled
 * cloud elm_deskew_downsample produced -- what the node publishes as its debug clouds (pcm.cpp:308-316). */
int elm_scan_download(const elm_scan* scan, float* xyz, size_t cap);

/* ---------------------------------------------------------------- registration -------------------- */
/* Registration::RunRegister (reg.hpp:122-124, reg.cpp:274-418; call sites pcm.cpp:280-282, 412-414) on host
 * buffers: uploads the scan, iterates on the device, downloads the result.  trace may be NULL or an array of
 * ELM_MAX_ITER_TRACE entries. */
int elm_register(elm_ctx* ctx, const elm_map* map, const float* scan_xyz, size_t n, const double T0[16],
                 const elm_reg_config* cfg, double T_out[16], int* is_success, double* fitness_score,
                 double local_cov[36], elm_reg_result* result, elm_iter_trace* trace);
/* What Registration::RunRegister writes to stdout for one call (reg.cpp:291-295, 343-356, 393-413), rebuilt from the call's result:
 * the warnings the reference prints whatever the configuration -- "VOXEL MAP EMPTY!", "[RunRegister] Small corresponding  ratio. r",
 * "[RunRegister] ICP Fitness Score Low f" -- and, with cfg->b_debug_print, its per-iteration line "[Registration] Total Correspondence
 * Time for: i in X ms, and cores num: N", the totals "[Registration] Total Correspondence Time: X ms" and "[Registration] RunRegister:
 * iteration N executed in Y ms", the corresponding ratio and the fitness score, with the reference's colour codes
 * (localization_functions.hpp:78-85) and stream formatting.  trace (per-iteration n_corr) and corr_ms (per-iteration correspondence
 * time in ms: the accumulate launch, which IS the correspondence search + the sums here) may be NULL: the per-iteration lines are then
 * left out.  Writes at most cap bytes including the terminating NUL; returns the length the whole text needs.  elm_register prints this
 * text itself (debug lines only with b_debug_print: the default adds no event, no trace and no output on success). */
size_t elm_format_register_log(const elm_reg_config* cfg, const elm_reg_result* res, size_t n_points, const elm_iter_trace* trace,
                               const double* corr_ms, double total_ms, char* buf, size_t cap);

/* The same on B device-resident scans against one map, all iterated together (one accumulate launch, one
 * optional all-reduce and one solve launch per ICP iteration for the whole batch).  T0: 16*B doubles.
 * results: B entries.  trace: NULL or B*ELM_MAX_ITER_TRACE entries. */
int elm_register_batch(elm_ctx* ctx, const elm_map* map, elm_scan* const* scans, int batch, const double* T0,
                       const elm_reg_config* cfg, elm_reg_result* results, elm_iter_trace* trace);
/* Continuous batching: `count` registrations through `slots` device slots.  Finished slots are refilled on the device with
 * the next pending registration after every ICP iteration, so every launch stays full until the queue is empty (throughput
 * mode for many more registrations than can usefully iterate in lockstep).  Results are bit-identical to
 * elm_register_batch on the same resident scans (which slot serves a registration may vary from call to call on one
 * rank -- the solve kernel hands out the queue positions -- but a registration's arithmetic does not depend on its slot; with a
 * communicator attached the assignment is in slot order, identical on every rank).  elm_register keeps the caller's point order
 * while elm_scan_upload orders the points: the two agree up to the order of the summation (1e-9 on every sum), not bit for bit.
 * trace: NULL or count*ELM_MAX_ITER_TRACE entries.  use_radar_cov = 1 with a covariance method: the registrations run as lockstep
 * batches of `slots` (same results as elm_register_batch; with a communicator attached the all-reduce then carries 64 sums per scan). */
int elm_register_stream(elm_ctx* ctx, const elm_map* map, elm_scan* const* scans, int count, const double* T0,
                        const elm_reg_config* cfg, int slots, elm_reg_result* results, elm_iter_trace* trace);
/* The same with the scans still in HOST memory when the call starts -- RunRegister's per-call contract (reg.cpp:274-290: the
 * caller hands over a point vector): scan_xyz[i] = packed float32 xyz of registration i, n_pts[i] points.  Uploads (DMA on a copy
 * stream, in groups of ~32 MB), the device-side ordering kernel and the ICP iterations of the registrations that have already
 * arrived overlap; a slot starts a registration as soon as its scan has landed.  Page-locked sources (elm_host_alloc or
 * hipHostRegister) are read by the DMA engines directly and reach the PCIe rate; pageable ones are staged by the runtime.  HBM
 * needed: every scan of the call (12 bytes per point) + THREE staging sets of one upload group (~32 MB of packed xyz) each + 4 bytes
 * per staged point of ordering scratch.  Results are bit-identical to elm_register_stream on
 * the same scans uploaded with elm_scan_upload.  One rank only: ELM_ERR_UNSUPPORTED with a communicator or hook attached. */
int elm_register_stream_host(elm_ctx* ctx, const elm_map* map, const float* const* scan_xyz, const uint32_t* n_pts, int count,
                             const double* T0, const elm_reg_config* cfg, int slots, elm_reg_result* results, elm_iter_trace* trace);
/* page-locked host memory for the sources of elm_register_stream_host / elm_scan_upload (NULL on failure) */
void* elm_host_alloc(size_t bytes);
void elm_host_free(void* p);
/* diagnostic: GB/s of one plain host-to-device copy of `bytes` from `host` on this box (median of reps) -- the PCIe rate a
 * host-fed stream sits under */
int elm_ctx_measure_h2d(elm_ctx* ctx, const void* host, size_t bytes, int reps, double* gb_per_s);
/* Asynchronous halves of elm_register_batch: enqueue everything on the context stream / wait and fetch results.  enqueue returns
 * without waiting for the device on the first batch of a context; afterwards it polls the device-side count of still-iterating
 * scans (one 4-byte read-back where the previous batch finished, then every second iteration) so that it can stop enqueueing
 * iterations early -- i.e. it may block for the iterations enqueued so far.  Exactly one batch may be in flight per context. */
int elm_register_batch_enqueue(elm_ctx* ctx, const elm_map* map, elm_scan* const* scans, int batch,
                               const double* T0, const elm_reg_config* cfg, int want_trace);
int elm_register_batch_finish(elm_ctx* ctx, elm_reg_result* results, elm_iter_trace* trace);

/* ---------------------------------------------------------------- deskew -------------------------- */
/* Tables produced by ImuDeskewInfo / OdomDeskewInfo (pcm.cpp:533-729). */
typedef struct elm_deskew_tables {
    double d_time_scan_cur;     /* pcm.cpp:474/481 */
    double d_time_scan_end;     /* pcm.cpp:475/480 */
    int32_t i_imu_pointer_cur;  /* last valid table index (pcm.cpp:580) */
    int32_t b_run_deskew;       /* loc.ini:86 */
    int32_t b_is_imu_available; /* pcm.cpp:584 */
    int32_t b_is_odom_available;/* pcm.cpp:728 */
    float f_odom_incre_x, f_odom_incre_y, f_odom_incre_z; /* pcm.cpp:725 */
    float _pad;
    const double* vec_d_imu_time;  /* [i_imu_pointer_cur + 1] host pointers */
    const double* vec_d_imu_rot_x;
    const double* vec_d_imu_rot_y;
    const double* vec_d_imu_rot_z;
} elm_deskew_tables;

/* The per-point loop of DeskewPointCloud (pcm.cpp:498-525 -> DeskewPoint :780-824) as one HIP kernel.
 * xyz: n*3 float32, rel_time: n float32 (already rebased as pcm.cpp:483-485), xyz_out: n*3 float32 (host).
 * Returns ELM_OK and *ok = 0 when IMU or odom tables are unavailable (pcm.cpp:494-496, nothing written). */
int elm_deskew(elm_ctx* ctx, const float* xyz, const float* rel_time, size_t n, const elm_deskew_tables* tab,
               float* xyz_out, int* ok);
/* The same per-point loop followed by VoxelHashMap::VoxelDownsample (vhm.hpp:260-283: the first point of every floor-keyed
 * voxel of edge voxel_size) fused on the device: the undistorted cloud never leaves HBM and comes back as a resident scan
 * (kept points in input order) for elm_register_batch; release it with elm_scan_destroy.  *ok as elm_deskew (*scan_out is
 * NULL when 0).  ELM_ERR_UNSUPPORTED when |coordinate / voxel_size| >= 2^20 (use elm_deskew + elm_voxel_downsample). */
int elm_deskew_downsample(elm_ctx* ctx, const float* xyz, const float* rel_time, size_t n, const elm_deskew_tables* tab,
                          double voxel_size, elm_scan** scan_out, int* ok);
/* Host-side table preparation, same arithmetic as the reference (doubles for the IMU table, float32
 * PCL/Eigen transforms for the odometry increment).
 * imu: n_imu rows (t, wx, wy, wz) already rotated into the ego frame (pcm.cpp:328).
 * odom: n_odom rows of 14 doubles (t, px,py,pz, qx,qy,qz,qw, vx,vy,vz, wx,wy,wz).
 * front_time/back_time: time field of the first/last raw point; stamp: message stamp - lidar_time_delay. */
int elm_deskew_prepare(const double* imu4, size_t n_imu, const double* odom14, size_t n_odom, double stamp,
                       float front_time, float back_time, int lidar_scan_time_end, int run_deskew,
                       double* tab_time, double* tab_rx, double* tab_ry, double* tab_rz, size_t tab_cap,
                       elm_deskew_tables* out);

/* ---------------------------------------------------------------- caller glue (host) -------------- */
/* The steps of PcmMatching::CallbackPointCloud / CallbackInitialPose either side of the device path (SURVEY.md 8
 * rows f2 / f4), in the reference's float32 / float64 arithmetic.  Host functions, no GPU needed. */
/* FilterPointsByDistance (pcm.cpp:451-465): drops points farther than max_dist (float norm). time may be NULL. */
int elm_filter_points_by_distance(const float* xyz, const float* time, size_t n, double max_dist, float* xyz_out,
                                  float* time_out, size_t* n_out);
/* VoxelHashMap::VoxelDownsample (vhm.hpp:260-283): index of the first point of every floor-keyed voxel, in input
 * order (the reference emits unordered_map order; only the set is contractual). */
int elm_voxel_downsample(const float* xyz, size_t n, double voxel_size, int64_t* keep_idx, size_t* n_keep);
/* GetInterpolatedPose (pcm.cpp:933-1045): odom rows as in elm_deskew_prepare; T_out = Eigen::Affine3f matrix,
 * column-major; *ok = 0 when no odometry at or before the time exists. */
int elm_get_interpolated_pose(const double* odom14, size_t n_odom, double d_cur_time, float T_out[16], int* ok);
/* Registration::CalFramePointCov / CalPointCov (registration.hpp:186-217; called at registration.cpp:302-305 under use_radar_cov): the
 * covariance term R S of every source point from its position (map frame under the initial guess at the call site) and the range /
 * azimuth / elevation spreads.  cov9: n column-major 3x3 (not symmetric).  Host arithmetic (glibc sin / cos / atan2); elm_register
 * evaluates the same formula inside its radar kernel with the device's math library: the two agree to a few ulp, not bit for bit, so a
 * RunRegister re-assembled from this call + GetCorrespondences* + AlignCloudsLocal* under use_radar_cov follows elm_register to the
 * tolerance of the sums (1e-9; an ill-conditioned first-iteration metric amplifies the difference), not to the last bit. */
int elm_cal_frame_point_cov(const double* xyz, size_t n, double range_var_m, double azim_var_deg, double ele_var_deg, double* cov9);

/* Covariance of the published odometry (PublishPcmOdom pcm.cpp:1082-1098, NormalizeCovariance pcm.hpp:248-268):
 * cov_out is the row-major 6x6 of nav_msgs/Odometry.pose.covariance. */
int elm_shape_odom_covariance(const double local_cov[36], const double icp_ego_pose[16], double d_icp_pose_std_m,
                              double cov_out[36]);

/* ---------------------------------------------------------------- CPU EKF (host) ------------------ */
/* Plain-CPU counterpart of the reference's 27-state EKF (ekf_localization/src/ekf_algorithm.cpp) -- SURVEY.md 8 row f1.
 * north_star keeps the filter on the CPU; these calls close the config-5 stream (ICP pose -> EKF update -> next seed).
 * Built: Init, RunPredictionImu (+ ComplementaryKalmanFilter), RunGnssUpdate (all sources incl. PCM / PCM_INIT),
 * GetCurrentState, and the node's CallbackPcmOdom / GnssTimeCompensation / state deque.  ZUPT, CAN update and IMU-mount
 * calibration (off in the shipped localization.ini) return ELM_ERR_UNSUPPORTED. */
typedef struct elm_ekf elm_ekf;
enum { ELM_GNSS_NOVATEL = 0, ELM_GNSS_NAVSATFIX = 1, ELM_GNSS_BESTPOS = 2, ELM_GNSS_PCM = 3, ELM_GNSS_PCM_INIT = 4 }; /* ls.hpp:28 */
typedef struct elm_ekf_config { /* [ekf_localization] keys of config/localization.ini:15-75 */
    double imu_gravity;
    int32_t imu_estimate_gravity, imu_estimate_calibration, use_zupt, use_complementary_filter, gps_type, _pad;
    double ekf_init_x_m, ekf_init_y_m, ekf_init_z_m, ekf_init_roll_deg, ekf_init_pitch_deg, ekf_init_yaw_deg;
    double state_std_pos_m, state_std_rot_deg, state_std_vel_mps, state_std_gyro_dps, state_std_acc_mps;
    double imu_std_gyro_dps, imu_std_acc_mps, ekf_imu_bias_cov_gyro, ekf_imu_bias_cov_acc;
    double gnss_min_cov_x_m, gnss_min_cov_y_m, gnss_min_cov_z_m, gnss_min_cov_roll_deg, gnss_min_cov_pitch_deg, gnss_min_cov_yaw_deg;
    double can_vel_scale_factor, ekf_can_meas_uncertainty_vel_mps, ekf_can_meas_uncertainty_yaw_rate_deg; /* RunCanUpdate */
} elm_ekf_config;
typedef struct elm_ekf_state { /* EkfState (ls.hpp) + covariance, state order of ekf_algorithm.hpp:41-69 */
    double x[27];            /* rotation slots (3..5, 24..26) are 0: the attitude lives in the quaternions */
    double rot_xyzw[4], imu_rot_xyzw[4];
    double P[27 * 27];       /* row-major */
    double timestamp;
    int32_t b_state_initialized, b_yaw_initialized, b_rotation_stabilized, b_state_stabilized, b_pcm_init_on_going, _pad;
} elm_ekf_state;
typedef struct elm_ego_state { /* the EgoState fields GetCurrentState fills (ekfa.cpp:778-833) */
    double timestamp, x_m, y_m, z_m, roll_rad, pitch_rad, yaw_rad, roll_vel, pitch_vel, yaw_vel, vx, vy, vz, ax, ay, az;
    double x_cov_m, y_cov_m, z_cov_m, roll_cov_rad, pitch_cov_rad, yaw_cov_rad;
} elm_ego_state;
void elm_ekf_config_default(elm_ekf_config* cfg);
int elm_ekf_create(const elm_ekf_config* cfg, elm_ekf** out);
void elm_ekf_destroy(elm_ekf* ekf);
/* RunPredictionImu (ekfa.cpp:167-316): gyro / acc already rotated into the ego frame (ImuStructConverter) */
int elm_ekf_predict_imu(elm_ekf* ekf, double timestamp, const double gyro[3], const double acc[3], int* predicted);
/* RunPrediction (ekfa.cpp:81-165): the constant-velocity model of use_imu = 0 (ekfl.cpp:204-216) */
int elm_ekf_predict(elm_ekf* ekf, double timestamp, int* predicted);
/* RunCanUpdate + ZuptCan (ekfa.cpp:434-506, 567-587): vehicle-frame velocity and angular rate (the node fills vel[0], gyro[2]) */
int elm_ekf_update_can(elm_ekf* ekf, double timestamp, const double vel[3], const double gyro[3], int* updated);
/* RunGnssUpdate (ekfa.cpp:318-432): pos_cov / rot_cov row-major 3x3 */
int elm_ekf_update_pose(elm_ekf* ekf, double timestamp, const double pos[3], const double quat_xyzw[4], const double pos_cov[9],
                        const double rot_cov[9], int source, int* updated);
/* CallbackPcmOdom / CallbackPcmInitOdom (ekfl.cpp:147-220): odometry pose + row-major 6x6 covariance, time-compensated
 * against the published state history (GnssTimeCompensation ekfl.cpp:323-394) */
int elm_ekf_update_pcm_odom(elm_ekf* ekf, double stamp, const double pos[3], const double quat_xyzw[4],
                            const double covariance36[36], int source, int* updated);
/* GeographicLib::LocalCartesian(ref).Forward (ekfl.cpp:643-648): WGS84 geodetic -> east / north / up metres at the reference point */
int elm_gps_project(double ref_lat_deg, double ref_lon_deg, double ref_alt_m, double lat_deg, double lon_deg, double alt_m, double xyz[3]);
/* CallbackNavsatFix (ekfl.cpp:92-125): position_covariance = the message's row-major 3x3 (its diagonal holds standard deviations, which
 * the node squares); use_gps / gnss_uncertainty_max_m = the [ekf_localization] keys use_gps / gnss_uncertainy_max_m */
int elm_ekf_update_navsatfix(elm_ekf* ekf, double stamp, double lat_deg, double lon_deg, double alt_m, const double position_covariance[9],
                             double ref_lat_deg, double ref_lon_deg, double ref_alt_m, int use_gps, double gnss_uncertainty_max_m,
                             double pos_out[3], int* updated);
int elm_ekf_get_state(elm_ekf* ekf, elm_ekf_state* out);
/* GetCurrentState + the state-history upkeep of PublishInThread (ekfl.cpp:397-410); call after every prediction */
int elm_ekf_publish(elm_ekf* ekf, elm_ego_state* out);

/* ---------------------------------------------------------------- formats (host) ------------------ */
/* On-disk / wire formats either side of the path so the reference's own map, localization.ini and
 * calibration.ini drive the drop-in. */
typedef struct elm_ini elm_ini;
/* IniParser::ParseConfig rules (bsw/system/ini_parser/ini_parser.cpp:41-225 over SimpleIni): getters return 1 = found,
 * 0 = key missing (output untouched), <0 = error.  Numbers use atoi/atof, so "5.0 ; comment" reads as 5.0. */
int elm_ini_load(const char* path, elm_ini** out);
void elm_ini_destroy(elm_ini* ini);
int elm_ini_get_string(const elm_ini* ini, const char* section, const char* key, char* buf, size_t cap);
int elm_ini_get_int(const elm_ini* ini, const char* section, const char* key, int* out);
int elm_ini_get_bool(const elm_ini* ini, const char* section, const char* key, int* out); /* atoi(v) > 0 */
int elm_ini_get_double(const elm_ini* ini, const char* section, const char* key, double* out);
int elm_ini_get_array(const elm_ini* ini, const char* section, const char* key, double* out, size_t cap, size_t* n);

typedef struct elm_pcm_node_config { /* PcmMatchingConfig fields the pipeline reads (pcm_matching_config.hpp; pcm.cpp:152-170) */
    char lidar_type[32];          /* "ouster" selects OusterCloudmsg2cloud */
    int32_t lidar_scan_time_end, pcm_voxel_max_point, run_deskew, input_index_sampling;
    double lidar_time_delay, pcm_voxel_size, input_max_dist, input_voxel_ds_m;
    double tf_ego_to_lidar[16];   /* column-major */
} elm_pcm_node_config;
void elm_pcm_node_config_default(elm_pcm_node_config* cfg);
/* ProcessINI (pcm.cpp:121-196): reads [common_variable] + [pcm_matching] from localization.ini and the "Rear To Main
 * LiDAR" / "Rear To Imu" rows of calibration.ini (ZYX Euler, lf.hpp:340-345).  Call the *_default functions first; keys
 * missing from the file leave their fields unchanged.  Either path may be NULL. */
int elm_load_pcm_config(const char* localization_ini, const char* calibration_ini, elm_pcm_node_config* node,
                        elm_reg_config* reg);
int elm_load_ekf_config(const char* localization_ini, elm_ekf_config* cfg); /* ekfl.cpp:250-316 */

/* pcl::io::loadPCDFile<PointXYZINormal> as used for the map (pcm.cpp:72-79): ascii / binary / binary_compressed PCD,
 * x y z (float32, matched by field name) -> freshly malloc'ed xyz[3n]; release with elm_free. */
int elm_pcd_load_xyz(const char* path, float** xyz_out, size_t* n_out);
void elm_free(void* p);

/* PointCloud2-style record unpack (pcm.hpp:81-106; pcm.cpp:900-930). */
enum { ELM_FIELD_UINT16 = 4, ELM_FIELD_UINT32 = 6, ELM_FIELD_FLOAT32 = 7 }; /* sensor_msgs/PointField datatypes */
typedef struct elm_cloud_field { char name[24]; uint32_t offset; int32_t datatype; } elm_cloud_field;
/* is_ouster = 0: PointXYZIT (x y z intensity time, float32).  is_ouster = 1: OusterPointXYZIRT -- every
 * index_sampling-th record, intensity = reflectivity, time = t * 1e-9f, and the output holds n/index_sampling + 1 slots
 * (a trailing default point when n is a multiple of index_sampling, as in the reference).  cap = capacity of the outputs
 * in points; intensity / rel_time may be NULL. */
int elm_scan_from_cloud(const void* data, size_t n_points, size_t point_step, const elm_cloud_field* fields, int n_fields,
                        int is_ouster, int index_sampling, float* xyz, float* intensity, float* rel_time, size_t cap,
                        size_t* n_out);

/* ---------------------------------------------------------------- the node callback ---------------- */
/* PcmMatching::CallbackPointCloud (pcm.cpp:198-324) as ONE call: stamp -= lidar_time_delay, FilterPointsByDistance,
 * DeskewPointCloud (tables on the host, per-point loop on the GPU), GetInterpolatedPose at the scan end, VoxelDownsample,
 * lidar pose = sync ego pose * tf_ego_to_lidar, RunRegister, ego pose = result * tf_ego_to_lidar^-1, covariance shaping.
 * imu4 / odom14: the node's deq_imu_ / deq_odom_ contents as in elm_deskew_prepare.  *published = 0 reproduces the
 * reference's silent returns (empty input, deskew data missing, no synced pose, registration failure: pcm.cpp:226-229,
 * 238-241, 249-251, 289-292); the other outputs are then undefined except result. */
typedef struct elm_pcm_scan_output {
    double pose_ego[16];     /* icp_ego_pose, column-major */
    double pose_lidar[16];   /* registration result, column-major */
    double covariance[36];   /* nav_msgs/Odometry.pose.covariance, row-major (PublishPcmOdom) */
    double fitness_score;
    double time_scan_end;    /* d_time_scan_end_ = stamp of the published odometry */
    uint64_t n_filtered;     /* points after the distance filter */
    uint64_t n_source;       /* points after VoxelDownsample = registration source size */
    elm_reg_result result;
} elm_pcm_scan_output;
int elm_pcm_callback_point_cloud(elm_ctx* ctx, const elm_map* map, const elm_pcm_node_config* node, const elm_reg_config* reg,
                                 const float* xyz, const float* point_time, size_t n, double stamp, const double* imu4,
                                 size_t n_imu, const double* odom14, size_t n_odom, elm_pcm_scan_output* out, int* published);

/* ---------------------------------------------------------------- multi-GPU ----------------------- */
/* (a) ONE process, N GPUs -- a device group (elm_ctx_create(device_ids[], n, &ctx); the reference's pcm_matching node is
 * one process that calls Registration::RunRegister, pcm.cpp:280-282).  elm_ctx_create_multi creates one context per entry of device_ids
 * and returns the first as the group's LEAD context; everything else keeps its signature.  On the lead:
 *   elm_map_build / elm_map_cal_voxel_cov_all / elm_map_cal_point_cov_all / elm_map_build_neighbourhoods   the map is REPLICATED on every
 *       device (the handle is rank 0's replica: read-backs, elm_map_get_correspondences, elm_map_find_ground_height work on it);
 *   elm_scan_upload          the scan is ordered along the ordering kernel's Hilbert curve and cut into N contiguous SHARDS, one per
 *       device (locality-aware sharding: a rank holds a compact sector of the scan at full density); elm_scan_size = the whole scan;
 *   elm_register             RunRegister on host buffers: the caller's point order cut into N contiguous shards;
 *   elm_register_batch / elm_register_stream   on scans uploaded through the lead;
 *   every ICP iteration all-reduces the ranks' packed normal equations (ONE ncclAllReduce(double, sum) of ELM_PACKED_SUMS doubles per
 *       scan over the communicators the ranks form among themselves -- RCCL over xGMI -- each rank driven by its own host thread) and
 *       every rank solves the same sums; the ranks' results are compared bit for bit (ELM_ERR_COMM if they differ);
 *   elm_ctx_destroy          destroys the group.
 * Not available on a group (ELM_ERR_UNSUPPORTED): elm_register_stream_host, elm_register_batch_enqueue / _finish.
 * elm_pcm_callback_point_cloud on a group takes its stage-by-stage path (deskew + downsample on the lead device, the registration
 * sharded): same results; the fused one-pass form is a plain context's (the callback registers ~10 k downsampled points).
 * A device id may repeat ({0, 0}: two ranks on one GPU).  RCCL refuses two ranks on one device; such a group exchanges through
 * page-locked host memory (sum in rank order) -- the form a one-GPU box can test.  ELM_GROUP_EXCHANGE=host | rccl forces either.
 * n = 1 returns a plain context (with ELM_GROUP_EXCHANGE=rccl: a group of ONE rank -- worker thread, one-rank communicator, one
 * ncclAllReduce per iteration: what a one-GPU box can run of the group's RCCL path). */
int elm_ctx_create_multi(const int* device_ids, int n, elm_ctx** out);
/* ranks of the group a context leads (1: a plain context), its exchange (0 none, 1 RCCL, 2 host memory), its devices */
int elm_ctx_group_info(elm_ctx* ctx, int* n_ranks, int* exchange, int* device_ids, int cap);
/* RunRegister on ONE RANK's shard of an n_total-point scan (host buffers, the caller's point order): what a rank of a process-per-GPU
 * job (b) calls where the one-GPU caller calls elm_register -- the sums are exchanged over the context's communicator / hook, the overlap
 * gate (reg.cpp:351) is taken against n_total.  quiet != 0: RunRegister's log text is not printed (one rank of a job prints it). */
int elm_register_shard(elm_ctx* ctx, const elm_map* map, const float* shard_xyz, size_t n, size_t n_total, const double T0[16],
                       const elm_reg_config* cfg, elm_reg_result* result, elm_iter_trace* trace, int quiet);

/* (b) One process per GPU.  Rank 0 obtains an id, the host distributes its bytes (e.g. torch.distributed
 * broadcast), every rank calls elm_comm_init.  Afterwards elm_register_batch* sums the packed normal
 * equations of every scan over all ranks with ONE ncclAllReduce(double, sum) per ICP iteration (RCCL/xGMI).
 * RCCL is dlopen'ed ("librccl.so.1") on first use, a single-GPU process never needs it. */
#define ELM_COMM_ID_BYTES 128
int elm_comm_get_unique_id(void* id_bytes /* ELM_COMM_ID_BYTES */);
int elm_comm_init(elm_ctx* ctx, int rank, int nranks, const void* id_bytes);
int elm_comm_destroy(elm_ctx* ctx);
/* rank / size as the RCCL communicator itself reports them (ncclCommUserRank / ncclCommCount); nranks = 0 without a communicator */
int elm_comm_info(elm_ctx* ctx, int* rank, int* nranks);
/* Alternative exchange hook (e.g. a torch.distributed all_reduce from Python): called between the accumulate
 * and the solve launches with the device pointer of the packed sums. Pass NULL to remove. */
typedef int (*elm_allreduce_fn)(void* dev_ptr, size_t n_doubles, void* hip_stream, void* user);
int elm_comm_set_hook(elm_ctx* ctx, elm_allreduce_fn fn, void* user);

/* number of doubles all-reduced per scan per iteration: 21 (upper JTJ) + 6 (JTr) + 1 (residual) + 1 (n_corr)
 * padded to 32 */
#define ELM_PACKED_SUMS 32

#ifdef __cplusplus
}
#endif
#endif /* ELIMALOC_HIP_H */
