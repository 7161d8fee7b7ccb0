// elm_internal.hpp -- device-side data layout shared by the kernels (elm_k_*.hip) and the C ABI (elm_api.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/elimaloc_hip.h"

namespace elm {

// ---- map in HBM -------------------------------------------------------------------------------------
// Open-addressing table keyed by the STORED voxel key (truncation toward zero, vhm.cpp:275).  One 32-byte slot
// carries everything a probe needs (key, voxel id, bucket range), so a hit costs one 32-B access.
struct __attribute__((aligned(32))) HashSlot {
    int32_t kx, ky, kz;
    int32_t vid; // -1 = empty
    uint32_t start;
    uint32_t cnt;
    uint32_t pad0, pad1;
};

struct __attribute__((packed, aligned(4))) Pt3 { // 12-byte point: scan points (one global_load_dwordx3), candidates of the neighbourhood lists
    float x, y, z;
};

struct __attribute__((aligned(16))) GridBlk { // four candidates of the cell grid, structure of arrays (48 bytes = three 16-byte loads)
    float x[4], y[4], z[4];
};

// One occupied neighbour voxel of a query voxel (VGICP / AVGICP lists), ONE 64-byte record = one memory sector per pair: the mean
// (float64, vhm.hpp:114-148), the plane normal and k of its regularised covariance U diag(1, 1, 1e-3) V^T = I - 0.999 n n^T, whose
// inverse is I + k n n^T with k = 999 (k = 0: the identity of voxels with fewer than two points) -- the 72-byte inverse covariance is
// rebuilt in registers instead of being fetched.  Maps with a covariance that is NOT of that form (rank-deficient neighbourhoods:
// round-off decides the signs of U against V, vhm.hpp:141) are found at map build and keep reading vox_cinv[vid].
struct __attribute__((aligned(64))) VoxRec {
    double mx, my, mz;
    double nx, ny, nz;
    double k;
    int32_t vid;
    int32_t pad; // position code of the neighbour (dx + 1) * 9 + (dy + 1) * 3 + (dz + 1)
};

// Four slots of a voxel-mean list (VGICP / AVGICP): the float32 means (the filter of the VGICP search) and, per slot, the voxel's id
// with the neighbour's position code (dx + 1) * 9 + (dy + 1) * 3 + (dz + 1) above it -- 64 bytes, one memory sector, never straddling a
// 128-byte line.  Padding slots: mean 1e18, word -1.  The float64 record of a voxel lives ONCE in DevMap::vox_rec.
constexpr uint32_t kVidBits = 26;                    // voxel ids of the lists: < 2^26 voxels (a 1.6 G-point map at 25 points per voxel)
constexpr uint32_t kVidMask = (1u << kVidBits) - 1u;
struct __attribute__((aligned(64))) VoxBlk {
    GridBlk g;
    int32_t vc[4];
};

struct DevMap {
    const HashSlot* slots;
    uint32_t mask; // capacity - 1 (capacity is a power of two >= 4 * n_voxels)
    uint32_t n_vox;
    uint32_t n_pts;
    uint32_t _pad;
    const float4* pts;      // bucket-ordered map points, xyz + pad (w unused), insertion order inside a bucket
    const double* vox_mean; // [n_vox][3]            CalVoxelCov (vhm.hpp:114-148)
    const double* vox_cov;  // [n_vox][9] row-major (read-backs: Covariances())
    const double* vox_cinv; // [n_vox][9] its inverse: what the VGICP / AVGICP pairs use (add_pair_world)
    const double* vox_nk;   // [n_vox][4] unit plane normal + k of the inverse's compact form I + k n n^T (see VoxRec)
    // GICP payload of a map point, ONE 128-byte record (a matched point costs one cache line instead of three scattered
    // ones): [0..2] neighbourhood mean, [3..11] INVERSE of the covariance of ProcessVoxelBlock (vhm.hpp:195-250), row-major,
    // [12..14] eigenvector of the covariance's smallest eigenvalue (reg.cpp:89-91, precomputed once), [15] pad
    const double* pt_gicp;  // [n_pts][16]
    const double* pt_cov;   // [n_pts][9] the covariance itself, row-major (read-backs; the use_radar_cov kernel inverts R^-1 C R^-T + C_source)
    double voxel_size;
    double inv_vs_exact; // 1 / voxel_size when voxel_size is a power of two (g * inv == g / voxel_size bit for bit), else 0
    // neighbourhood lists (optional): for every FLOOR-keyed query voxel that has at least one stored neighbour, the
    // points of its 27 trunc-keyed neighbour buckets concatenated in the reference's visiting order (x-major ..
    // z-minor, insertion order inside a bucket).  27x duplication of the map points, laid out for streaming.
    const HashSlot* qslots; // key = query voxel, vid = query id, start/cnt = its list, pad0 = occupied neighbours
    uint32_t qmask;
    uint32_t n_q;
    const Pt3* nbr_pts;       // candidate coordinates, 12 bytes each
    const uint32_t* nbr_idx;  // global point index of every candidate (read only for a GICP winner)
    // voxel-mean lists (VGICP, optional): the occupied voxels among the 27 neighbours of every query voxel in the
    // reference's visiting order (vhm.cpp:208-243), 32-byte records; vqslots: query key -> (start, cnt)
    const HashSlot* vqslots;
    uint32_t vqmask;
    uint32_t _pad2;
    const VoxRec* vox_rec;    // [n_vox] the float64 record {mean, normal, k, vid} of every voxel, ONCE (round 6; rounds 2-5 replicated it into
                              // every query list: 27 copies, 4.3 GB on the 50 M-point map)
    const uint32_t* vq_dense; // optional: the same (start << 5 | cnt) addressed by the dense floor-key box (vq_x0.., no hash probe)
    const VoxBlk* vnbr_blk;     // the lists: blocks of four slots {float32 mean, vid | code}: list q starts at block
                                // start(q) / 4 (list starts are multiples of four slots), padding slots 1e18 / -1
    uint32_t vnbr_pad_blk;      // index of the all-padding block at the end of vnbr_blk
    const VoxRec* vface;        // optional (with vq_dense): the FACE neighbours (+ the voxel itself) of every query voxel, in list order --
                                // what AVGICP pairs with (vhm.cpp:153-206) -- and
    const uint32_t* vqf_dense;  // (start << 3 | count) of those, addressed like vq_dense
    int32_t vq_x0, vq_y0, vq_z0;
    int32_t vq_nx, vq_ny, vq_nz;
    const uint16_t* nbr_cell_off; // [n_q][224]: every list is sorted by half-voxel cell (6x6x6 grid, clamped); cell c =
                                  // entries [off[c], off[c+1]) of the list
    // dense half-voxel cell grid over the map's bounding box (optional; default search index when it fits the memory budget):
    // the map points ONCE, sorted by cell (x-major, y, z fastest -- the cells iz0..iz1 of one (ix, iy) column are one contiguous
    // run), in 48-byte blocks of four (x[4], y[4], z[4]); every cell is padded to whole blocks with far-away points (1e18), so a
    // block never straddles cells and no candidate needs masking.  grid_start[linear cell] = first BLOCK of the cell, the next
    // entry = its end.  Candidate id = block * 4 + slot.  Cells follow the
    // STORED (truncated) voxel keys: bucket k > 0 is cells {2k, 2k+1}, bucket 0 (two voxels wide) is {-2,-1,0,1}, bucket k < 0 is
    // {2k-2, 2k-1}; geometrically cell c is [c h, (c+1) h], h = voxel_size / 2.
    const GridBlk* grid_blk;    // [n_blk]; block 0 is all padding (the target of masked-off loads), cells start at block 1
    const uint32_t* grid_idx;   // [4 * n_blk] bucket-order index of every candidate slot (GICP payload, insertion order for exact
                                // ties); 0xFFFFFFFF in padding slots
    const uint32_t* grid_start; // [gnx * gny * gnz + 4]
    // two-level form of the same grid (maps whose bounding box is too large or too sparse for one dense offset table): the box is cut
    // into tiles of kTile x kTile (x, y) cells; grid_tiles[tile] = {first offset entry, z0 | nz << 16}: the tile stores offsets only for
    // the cells z0 .. z0 + nz - 1 that hold points, (nz + 1) entries per column (its cell starts + the column end), columns (x, y)-major.
    // Blocks are sorted (tile, column, z), so the cells of a column are still one contiguous run.  Empty tiles share 64 zero entries.
    const uint2* grid_tiles;    // [gnx / kTile][gny / kTile] (gnx, gny are multiples of kTile)
    int32_t grid_tiled;         // 1: grid_start is addressed through grid_tiles
    int32_t gtny;               // tiles along y
    int32_t grid_wide;          // 1: the block array is 4 GB or more -- stage 1 addresses it in 16-byte units (template flag WIDE)
    uint32_t grid_nslots;       // 4 * n_blk: candidate slots of the block array
    const double* grid_gicp;    // [4 * n_blk][16]: pt_gicp gathered into slot order -- a GICP match reads its record without the index hop
                                // (only for maps with a covariance outside the compact form)
    const double* grid_gicp8;   // [4 * n_blk][8]: the compact record {mean[3], unit normal[3], k, -}: 64 bytes = one memory sector per match;
                                // the inverse covariance I + k n n^T is rebuilt in registers (k = 999: U diag(1, 1, 1e-3) V^T of
                                // vhm.hpp:238-246 with U = V up to rounding, checked per point at map build; k = 0: identity)
    int32_t gicp_compact;       // 1: grid_gicp8 is what the grid kernel reads (k = NaN flags a point outside the compact form: its full
                                // record is read from pt_gicp by the index in word 7); 2: the same and no point is flagged -- the kernel
                                // instantiation without the fallback, its pair gathered fused (pair_sum_compact)
    int32_t vox_compact;        // 1: the VoxRec's own normal / k are used (k = NaN: vox_cinv[vid] is read for that voxel); 2: no voxel flagged
    int32_t vface_plain;        // 1: the face sublists are written for the fused AVGICP walk (identity covariances as zero normals), which gathers
                                // sum w and sum (w k) n n^T instead of nine entries per pair
    int32_t vface_flagged;      // 1: some voxel of this map is outside the compact form (its records carry NaN in the normal): the fused walk
                                // skips those pairs and a fix-up launch adds them (RegParams::flagged)
    int32_t gx0, gy0, gz0;      // cell coordinates of grid entry (0, 0, 0)
    int32_t gnx, gny, gnz;
    // dense voxel box of the floor keys a query can have near the map: cnt27 | nocc27 << 16 of the reference's 27-voxel walk
    // (the work counters, and cnt27 == 0 -> no neighbour bucket at all -> the reference's origin default, vhm.cpp:37)
    const uint32_t* vox_stat;   // [vnx * vny * vnz]
    int32_t vx0, vy0, vz0;
    int32_t vnx, vny, vnz;
};

__host__ __device__ __forceinline__ uint32_t hash3(int32_t x, int32_t y, int32_t z) {
    uint32_t h = (uint32_t)x * 73856093u ^ (uint32_t)y * 19349669u ^ (uint32_t)z * 83492791u;
    h ^= h >> 16;
    h *= 0x7feb352du;
    h ^= h >> 15;
    h *= 0x846ca68bu;
    h ^= h >> 16;
    return h;
}

// ---- scans / per-registration state -----------------------------------------------------------------
struct ScanDesc {
    const Pt3* pts;    // sensor-frame points, packed xyz (12 bytes: what the LiDAR driver / the caller holds, no repack on upload)
    uint32_t n;        // points resident on this GPU
    uint32_t n_total;  // points of the whole scan (== n on one GPU)
    uint32_t blk_begin; // first logical block of this scan in the batch launch
    uint32_t blk_end;
};

// one per scan of the batch, lives in HBM for the whole registration
struct ScanState {
    double T[16];    // current pose, column-major
    double Rinv[9];  // row-major inverse of the rotation block (3x3 cofactor inverse, reg.cpp:79)
    double tinv[3];  // -Rinv t
    double fitness;  // d_fitness_score_
    double local_cov[36]; // column-major == row-major (symmetric)
    double n_corr_last;
    double pt_iters, cand_total, occ_total, fallback_blocks, tested_total; // work counters over the executed iterations
    int32_t done;
    int32_t success;
    int32_t gate;
    int32_t iters;
    int32_t reg;  // registration this slot works on (== slot index in a lockstep batch; -1 = idle slot of a stream)
    int32_t _pad;
};

// continuous batching (elm_register_stream): pending registrations and the device-side bookkeeping
struct QueueItem {
    const Pt3* pts;
    uint32_t n, n_total;
};
struct StreamCtrl {
    int32_t next;      // first registration not yet assigned to a slot
    int32_t completed; // registrations whose final state has been saved
    int32_t total;
    int32_t ready;     // host-fed streams: registrations whose scan has arrived in HBM (uploaded + ordered); a slot only takes r < ready
    int32_t done_iter; // iteration (0-based) whose solve finished the LAST registration, -1 while the stream runs: what the next call of the
                       // same shape enqueues before it first looks (the host's own count only says when it LOOKED)
};

// in-solve refill of finished slots (single-rank streams, see finish_slot in elm_k_solve.hip)
struct StreamArgs {
    ScanDesc* scans;
    const QueueItem* queue;
    const double* qT0;
    ScanState* out_state;
    StreamCtrl* ctrl; // nullptr: no refill in the solve
    int32_t hostfed;  // 1: the queue fills while the stream runs (ctrl->ready grows): idle slots look for work at every solve
    int32_t save_only; // 1 (multi-rank streams with the refill launch): the solve saves a finished registration's state and counts it, the slot
                       // stays free for k_stream_refill to hand it the next registration in slot order
    int32_t stride;   // > 0 (multi-rank streams): slot s serves the registrations s, s + stride, s + 2 stride, ... -- an assignment that
                      // is a function of the slot alone, hence identical on every rank without any exchange; 0: first come, first served
    int32_t iter;     // index of this iteration in the call (StreamCtrl::done_iter)
};

struct RegParams {
    double th;  // max_search_dist
    double th2; // th * th
    double lm_lambda;
    double term_thr;
    double min_overlap;
    double max_fitness;
    int32_t method;
    int32_t max_iter;
    uint32_t uniform_blocks; // > 0: every scan / slot owns exactly that many consecutive workgroups (scan = block / uniform_blocks)
    int32_t radar;           // use_radar_cov with a covariance method: k_accumulate_radar's 64-double partial records, full JTJ
    int32_t _reserved0;
    int32_t stats;           // elm_ctx_set_work_counters: the grid / voxel-list kernels also sum the three work counters (STATS = 1)
    uint32_t* flagged;   // [workgroups] AVGICP on a map with flagged voxels: set by a workgroup of the fused walk that met (and skipped) a flagged
                         // record, read and cleared by the fix-up launch that adds those pairs to the workgroup's partial record (nullptr: the
                         // map runs the nine-entry walk with its fallback instead)
    int32_t rank_check;  // 1 (several ranks, production kernels): slots 29..31 of every scan's exchanged sums -- the work counters, zero in
                         // production -- carry (1, id, id^2), id = 16 (registration + 1) + (iteration & 15): after the all-reduce every rank
                         // verifies sum(id) == n id and sum(id^2) == n id^2, i.e. that all ranks iterate the SAME registration in this slot
                         // (the refill launch hands registrations out from flags every rank computes for itself); a mismatch sets active[1]
                         // and the call returns ELM_ERR_COMM instead of adding up unrelated normal equations
    int32_t _pad_rc;
    double* asym;        // [workgroups][16] side records of a map with an asymmetric flagged covariance (the strict lower triangle of
                         // H_w - H_w^T, see asym_side_store in elm_dev_reduce.hpp): written by every workgroup of such a launch, reduced by
                         // k_solve, which restores all 36 entries of J^T M J before the congruence; nullptr on every other map
    double* asym_sums;   // [scans][16] the scans' reduced side sums when the iteration is split around an exchange (modes 1 / 2 of k_solve;
                         // laid out right behind the packed sums, so ONE all-reduce carries both)
    double radar_var[3]; // range_variance_m, azimuth_variance_deg, elevation_variance_deg (reg.hpp:77-79)
    // correspondence queries (elm_map_get_correspondences: the reference's public VoxelHashMap::GetCorrespondence* calls, vhm.cpp:31-206):
    // the QUERY instantiations of the grid / voxel-list kernels (template STATS = 2) take point i of their one "scan" from query[3 i ..]
    // (GLOBAL frame, float64: what Registration hands the map after TransformPoints) instead of T * p, run the production search and
    // write the pair(s) instead of sums: q_out[i] = the map point (bucket order) / voxel id of the pair, -1 = the reference's default
    // target at the origin (no neighbour bucket at all, vhm.cpp:37 / :105), -2 = no pair (beyond max_dist); GetCorrespondencesAllCov:
    // q_out[8 i + r] = voxel id of the pair with the r-th voxel of the reference's visiting order (vhm.cpp:224-230), -2 = none
    const double* query;
    int32_t* q_out;
};
constexpr int kAsymRecord = 16; // doubles per side record (15 used)
constexpr int kRadarRecord = 64; // doubles per partial record of k_accumulate_radar

constexpr int kTileShift = 3, kTile = 1 << kTileShift; // two-level grid: tiles of 8 x 8 columns
constexpr int kBlock = 256;
constexpr int kInitPack = 8; // batches up to this size are initialised from kernel arguments (k_init_pack)
struct InitPack {
    ScanDesc d[kInitPack];
    double T0[kInitPack][16];
};
constexpr int kSums = ELM_PACKED_SUMS; // 21 + 6 + 1 + 1 (+ n_cand, n_occ, pad)

// ---- launchers (elm_k_*.hip) ----------------------------------------------------------------------
void launch_init_state(hipStream_t s, ScanState* st, const double* T0, int batch, int map_empty, int* active);
void launch_init_pack(hipStream_t s, ScanDesc* scans, ScanState* st, const InitPack& pack, int batch, int map_empty, int* active, const unsigned* n_dev);
void launch_stream_refill(hipStream_t s, ScanDesc* scans, ScanState* st, int slots, const QueueItem* queue, const double* qT0,
                          /* first: initial fill; save: copy finished states out + count them (0: the solve has done that) */
                          ScanState* out_state, StreamCtrl* ctrl, int first, int save = 1);
void launch_accumulate_radar(hipStream_t s, const DevMap& m, const ScanDesc* scans, int batch, int total_blocks, ScanState* st, double* partials,
                             const RegParams& rp); // use_radar_cov = 1, methods GICP / VGICP / AVGICP
void launch_accumulate_direct(hipStream_t s, const DevMap& m, const ScanDesc* scans, int batch, int total_blocks,
                              ScanState* st, double* partials, const RegParams& rp);
// mode 0: reduce + solve (single GPU); 1: reduce only -> sums; 2: solve only from sums
void launch_solve(hipStream_t s, const ScanDesc* scans, int batch, ScanState* st, const double* partials,
                  double* sums, const RegParams& rp, elm_iter_trace* trace, int mode, int* active, const StreamArgs* refill = nullptr);
void launch_nbr_count(hipStream_t s, const DevMap& m, const int32_t* qkeys, uint32_t n_q, uint32_t* counts, uint32_t* nocc);
void launch_accumulate_vnbr(hipStream_t s, const DevMap& m, const ScanDesc* scans, int batch, int total_blocks,
                            ScanState* st, double* partials, const RegParams& rp);
void launch_vnbr_fill(hipStream_t s, const DevMap& m, const int32_t* qkeys, uint32_t n_q, const uint32_t* offsets, VoxBlk* out_blk);
void launch_vox_rec_fill(hipStream_t s, const DevMap& m, VoxRec* out);
// face-neighbour sublists of the voxel-mean lists: counts (out == nullptr) or the records at face_off
void launch_vface(hipStream_t s, const DevMap& m, const uint32_t* offsets, const uint32_t* counts, uint32_t n_q, uint32_t* face_cnt,
                  const uint32_t* face_off, VoxRec* out, int plain);
void launch_accumulate_cell(hipStream_t s, const DevMap& m, const ScanDesc* scans, int batch, int total_blocks,
                            ScanState* st, double* partials, const RegParams& rp);
int stream_max_slots(); // slots one elm_register_stream call can iterate concurrently
void launch_accumulate_grid(hipStream_t s, const DevMap& m, const ScanDesc* scans, int batch, int total_blocks,
                            ScanState* st, double* partials, const RegParams& rp);
// Registration::AlignCloudsLocal / AlignCloudsLocalPointCov / AlignCloudsLocalVoxelCov (reg.cpp:15-225) on explicit pairs (elm_align_clouds_local)
struct AlignArgs {
    double Rinv[9], tinv[3]; // inverse of last_icp_pose: rotation block (row-major) and -Rinv t
    double th, th2;          // trans_th
    double lm_lambda;
    int32_t method;          // ELM_P2P / ELM_GICP / ELM_VGICP (= AVGICP)
    int32_t use_src_cov;     // use_radar_cov: the source points' covariance terms are added (reg.cpp:109-111 / 188-190)
};
constexpr int kAlignOut = 16 + 36 + 1 + 6 + 36 + 6 + 1; // T (column-major), local_cov (row-major), fitness, x, J^T M J (row-major), J^T M r, pair count
void launch_align_pairs(hipStream_t s, const double* src_local, const double* tgt_xyz, const double* tgt_cov, const double* src_cov, size_t n,
                        const AlignArgs& a, double* partials, double* out);
void launch_query_direct(hipStream_t s, const DevMap& m, int what, const double* query, size_t n, double th2, int32_t* q_out); // the plain walk (27 / 7 hash probes, float64): the query form of k_accumulate_direct
void launch_vox_stat(hipStream_t s, const DevMap& m, uint32_t* out); // fills DevMap::vox_stat's box (m.vx0.., m.vnx..)
void launch_gather_gicp(hipStream_t s, const DevMap& m, size_t n_slots, double* out, int compact); // pt_gicp[grid_idx[slot]] -> out[slot] (16 or 8 doubles)
// half-voxel cell of a stored coordinate (host + device; the binning of DevMap::grid_pts)
__host__ __device__ inline int grid_cell_of(double a, double voxel_size) {
    const double t = a / voxel_size; // the reference's own key arithmetic (vhm.cpp:275), truncated below
    const int k = (int)t;
    if (k > 0) return 2 * k + ((t - (double)k >= 0.5) ? 1 : 0);
    if (k < 0) return 2 * k - 2 + ((t - (double)k > -0.5) ? 1 : 0);
    if (t >= 0.0) return (t >= 0.5) ? 1 : 0;
    return (t > -0.5) ? -1 : -2;
}
void launch_nbr_cellsort(hipStream_t s, const DevMap& m, const int32_t* qkeys, uint32_t n_q, const uint32_t* offsets,
                         const uint32_t* counts, Pt3* pts, uint32_t* idx, uint16_t* cell_off);
size_t nbr_cell_stride();
void launch_nbr_fill(hipStream_t s, const DevMap& m, const int32_t* qkeys, uint32_t n_q, const uint32_t* offsets, Pt3* out,
                     uint32_t* out_idx);
// *bad counts the covariances whose inverse is not I + k n n^T to 1e-10 relative (the compact records are then not used)
void launch_voxel_cov(hipStream_t s, const DevMap& m, const uint2* ranges, double* vox_mean, double* vox_cov, double* vox_cinv, double* vox_nk, unsigned* bad);
void launch_point_cov(hipStream_t s, const DevMap& m, double d2max, double* pt_gicp, double* pt_cov, unsigned* bad); // pt_cov [n_pts][9]: read-backs

struct DeskewDev {
    double time_scan_cur, time_scan_end;
    int32_t imu_pointer_cur;
    int32_t odom_available;
    float incre_x, incre_y, incre_z;
    float _pad;
    const double* imu_time;
    const double* rot_x;
    const double* rot_y;
    const double* rot_z;
};
// ordered device-side VoxelDownsample: table / first sized 2^cap_log2 (table = all ones, first = all ones before the call),
// slot n entries, block_count ceil(n / 1024) entries; *total = kept points, *overflow = 1 when a key does not pack
void launch_voxel_downsample(hipStream_t s, const float* xyz, uint32_t n, double vs, unsigned long long* table, unsigned* first,
                             unsigned cap_log2, unsigned* slot, unsigned* block_count, unsigned* total, int* overflow, Pt3* out);
// Scan ordering on the device (k_scan_order): one workgroup per scan sorts its points along a Hilbert curve over 2 m sensor-frame
// cells (deterministic counting sort, no atomics on the data path).  jobs[j] = {src, dst, n, tmp}; src == dst is not allowed.
struct OrderJob {
    const Pt3* src; // caller's order
    Pt3* dst;       // Hilbert order
    uint32_t* tmp;  // n words of scratch
    uint32_t n;
    uint32_t _pad;
};
void launch_scan_order(hipStream_t s, const OrderJob* jobs, int n_jobs, const uint16_t* hilbert_lut);
// one scan ordered by many workgroups (three launches, the same bytes as k_scan_order); scratch = order_wide_scratch_bytes(n) bytes
unsigned order_wide_groups(unsigned n);
size_t order_wide_scratch_bytes(unsigned n);
void launch_scan_order_wide(hipStream_t s, const OrderJob* job, unsigned n, const uint16_t* hilbert_lut, void* scratch);
void launch_publish_ready(hipStream_t s, StreamCtrl* ctrl, int ready);
void launch_slots_idle(hipStream_t s, ScanDesc* scans, ScanState* st, int slots, unsigned cap_blocks);
constexpr int kOrderCells = 64; // cells per axis of the ordering grid (2 m cells: +-64 m around the sensor, clamped beyond)
void launch_deskew(hipStream_t s, const float* xyz, const float* rel_time, uint32_t n, const DeskewDev& d,
                   float* xyz_out);

} // namespace elm
