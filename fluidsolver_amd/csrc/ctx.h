// ctx.h -- private layout of the solver context (shared by context.hip and slab.hip).
//
// One context = one ROW SLAB of a W x H solver: global rows [r0, r1) owned, plus up to G ghost
// rows on each interior side held in the same buffers (rows [b0, b1)).  A single GPU is the
// one-slab case: r0 = b0 = 0, r1 = b1 = H, G = 0, no communicator.
#pragma once
#include "common.h"
#include <functional>
#include <unordered_map>
#include <vector>

namespace fs {

enum { F_VEL = 0, F_DYE, F_P, F_DIV, F_VORT, F_COUNT };
constexpr int kInf = 1 << 30;

// A field and its ping-pong partner (the reference swaps texture references, Solver:516,:540,:558,
// :572,:584).  `valid` = how many ghost rows beside the owned rows still equal the neighbour's
// owned rows (kInf: everywhere, e.g. an all-zero field).
// The velocity of a slab has a THIRD buffer (spr): a frame that speculates on its advection halo is checked only after
// the next frame's stencil chain has been enqueued, and until then the velocity its advection read must survive, so
// that chain writes into the spare instead (rotate(), see run_frames in slab.hip).
struct Field {
    float *buf[3] = {nullptr, nullptr, nullptr};
    int cur = 0, oth = 1, spr = 2;
    int np = 1;
    int valid = kInf;
    float *p() const { return buf[cur]; }
    float *t() const { return buf[oth]; }
    void flip() { const int k = cur; cur = oth; oth = k; }
    void rotate() { if (buf[spr]) { const int k = oth; oth = spr; spr = k; } }
};

struct Xfer {
    int field, bufidx, depth;   // refresh the ghost rows of f[field].buf[bufidx] to `depth` rows
};

enum { STEP_COMPUTE = 0, STEP_EXCHANGE = 1, STEP_CFL_REDUCE = 2, STEP_JOIN = 3, STEP_TRACERS = 4 };
// An exchange runs on the context's exchange stream.  Its completion event is a "slot"; the main
// stream joins a slot (waits for the event) before the first kernel that needs the rows.
enum { SLOT_VEL = 0, SLOT_DYE = 1, SLOT_SYNC = 2, SLOT_COUNT = 3 };

// One entry of a frame plan.  The plan of a frame is the same on every rank (same kinds, same
// depths), which is what lets an in-process group run K slabs in lock step.
struct Step {
    int kind = STEP_COMPUTE;
    const char *what = "";
    bool hidden = false;                // stream plumbing (fork / join of a side chain): not part of the plan dump
    int yb = 0, ye = 0, n = 0;          // rows computed / sweeps (plan dumps, tests)
    std::function<int()> run;           // STEP_COMPUTE: enqueue the kernel(s) on the context's stream
    std::vector<Xfer> xf;               // STEP_EXCHANGE
    int slot = SLOT_SYNC;               // STEP_EXCHANGE / STEP_JOIN
    bool async = false;                 // STEP_EXCHANGE: do not join right away (a STEP_JOIN follows later)
};

void slab_rows(int H, int world, int rank, int *r0, int *r1);

// p + n, except that a plan-only context has no buffers: null stays null (null + n is undefined behaviour)
template <class T>
inline T *off(T *p, size_t n) { return p ? p + n : p; }

}  // namespace fs

// Transport of the halo exchange.  kind 1 = RCCL (one process per GPU, ncclSend/ncclRecv over xGMI).
// kinds 0 and 2 = in-process groups (all slabs live in this process) used to test the slab logic on a
// 1-GPU box: kind 0 copies with blocking device memcpys between full synchronisations (the simplest
// possible transport); kind 2 copies asynchronously on the exchange streams, ordered only by the
// same events the RCCL path uses, so the K slabs really run concurrently and a missing dependency
// shows up as a wrong result.
// kind 3 = loopback, a measurement aid: ONE rank of a `world`-rank run whose peers do not exist.  Every transfer is a device
// copy of the same size out of the slab's own rows, the CFL reduction returns the slab's own bound: the fields are
// meaningless, but the slab runs the plan, the launches, the copies and the host work of a real rank, alone on its GPU.
struct fs_comm {
    int kind = 0;
    int world = 1;
    int rank = 0;                       // RCCL, loopback
    int device = 0;                     // RCCL only
    void *nccl = nullptr;               // ncclComm_t
    std::vector<fs_ctx *> members;      // local group: index = rank
    bool solo() const { return kind == 1 || kind == 3; }   // one rank per communicator object (stepped by fs_step)
    // measurement only (FS_OPT_SLAB_SERIALIZE, local groups): the kernels of the K slabs, which share one GPU here, run one
    // slab after the other instead of side by side -- K x the time one slab's frame takes on a GPU of its own
    int serialize = 0;
    hipEvent_t last_serial = nullptr;   // end of the launches most recently enqueued by any member
};

struct fs_ctx {
    int W = 0, H = 0, pitch = 0, device = 0;
    // slab geometry
    fs_comm *comm = nullptr;
    int world = 1, rank = 0;
    int r0 = 0, r1 = 0;                 // owned rows
    int b0 = 0, b1 = 0;                 // rows held by the buffers
    int G = 0;                          // ghost capacity per side
    bool dry = false;                   // plan-only context (no device memory), fs_slab_plan_dump
    size_t plane = 0;                   // floats per plane = pitch * (b1 - b0)
    fs_grid grid{};                     // owned rows (the whole grid on one GPU)
    hipStream_t stream = nullptr;
    fs::Field f[fs::F_COUNT];
    float *display = nullptr;           // 4 planes
    int *err_flag = nullptr;            // device
    // CFL bound, two slots used by alternate frames (the host reads frame f's value while frame f + 1 already reduces its own)
    int cfl_slot = 0;
    float *cfl_dev = nullptr;           // device, 2 floats: max |v_y| of the owned rows
    float *cfl_host = nullptr;          // pinned, 2 x `world` floats (RCCL uses the first of each slot)
    float cfl_max = 0.0f;               // all-reduced value of the current frame
    // exchange stream: every halo transfer and the CFL reduction run here, next to the kernels
    hipStream_t xstream = nullptr;
    hipEvent_t ev_main = nullptr;       // main stream: "rows are final" (what a transfer waits for)
    bool ev_main_fresh = false;         // ... and nothing has been enqueued on the main stream since it was last recorded
    hipEvent_t ev_copied[fs::SLOT_COUNT] = {nullptr, nullptr, nullptr};   // kind 2: my receives are done
    hipEvent_t ev_slot[fs::SLOT_COUNT] = {nullptr, nullptr, nullptr};     // transfer complete (sends and receives)
    bool pending[fs::SLOT_COUNT] = {false, false, false};
    hipEvent_t ev_cfl[2] = {nullptr, nullptr};   // CFL value of the slot has landed in cfl_host
    hipEvent_t ev_serial = nullptr;     // FS_OPT_SLAB_SERIALIZE: my launches so far are done
    int overlap = 1;                    // FS_OPT_SLAB_OVERLAP: speculative halo + split advection
    // FS_OPT_SLAB_PROFILE: timed event pairs around every join, so a run can report how long the kernels really
    // waited for halo rows (the part of the traffic that was NOT hidden)
    int profile = 0;
    struct WaitProbe { hipEvent_t a, b; int slot; };
    std::vector<WaitProbe> probes_live, probes_free;
    double wait_ms[fs::SLOT_COUNT] = {0.0, 0.0, 0.0};
    long wait_count[fs::SLOT_COUNT] = {0, 0, 0};
    int halo_hist[4] = {-1, -1, -1, -1};   // advection halos the last frames really needed (-1: unknown), newest first
    fs_params params{};
    int use_graph = 1;
    int stable_frames = 0;              // frames since the parameters / options last changed (graphs are recorded once they settle)
    int with_particles = 0;
    int fuse = 1;
    fs::JacobiTuning tuning{};          // block depths / tile shapes of this context's Jacobi loops (starts from the process default)
    // independent planes of a diffuse loop as two concurrent chains (second stream, fork/join by event)
    int split_planes = 1;
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // the projection of a single-GPU frame as two row bands on stream / stream2 (project_n_bands): band B iterates in
    // two private planes, so after such a frame the rows [p_band_row, H) of the pressure field live in
    // p_band[p_band_final] until settle_pressure() copies them home (only a read-back of the pressure asks for that)
    int split_rows = 1;                 // FS_OPT_SPLIT_ROWS
    float *p_band[2] = {nullptr, nullptr};
    hipEvent_t ev_band[2] = {nullptr, nullptr};
    bool p_band_active = false;         // planner: the current plan projects in two bands
    bool p_band_dirty = false;          // a frame has run since the rows were last copied home
    int p_band_row = 0, p_band_final = 0;
    // the dye diffusion loop of a single-GPU frame runs beside the whole velocity chain on a third stream
    hipStream_t stream3 = nullptr;
    hipEvent_t ev_fork3 = nullptr, ev_join3 = nullptr;
    bool p_zero = false;                // planner state: pressure iterate is implicitly 0
    bool vort_stale = false;            // the last frame computed curl + confinement in one pass: the vorticity plane was not written
    std::vector<fs::Step> steps;
    std::unordered_map<unsigned long long, hipGraphExec_t> graphs;
    std::unordered_map<unsigned long long, unsigned long long> graphs_after;  // ping-pong state a replay leaves
    std::unordered_map<unsigned long long, unsigned long long> graphs_epoch;  // param_epoch the graph's kernel arguments are from
    struct Splat { float px, py, fx, fy, r, g, b, radius, dt; };
    // Splats of a single-GPU context whose frames are replayed are not launched by fs_add_force: they wait here (at most
    // kMaxPendingSplats) and become the first nodes of the next fs_step's graph, their arguments set per replay; any other
    // call that looks at or enqueues on the fields launches them first (flush_splats).
    static constexpr int kMaxPendingSplats = 8;
    int defer_splats = 0;               // FS_OPT_SPLAT_IN_GRAPH (off: measured slower on this runtime, profiles/r03_o_splat_in_graph.log)
    std::vector<Splat> pending_splats;
    std::unordered_map<unsigned long long, hipGraph_t> graphs_src;                    // the graph an exec was instantiated from (owns the splat nodes)
    std::unordered_map<unsigned long long, std::vector<hipGraphNode_t>> graphs_splat_nodes;
    std::vector<fs::SplatNode> splat_scratch;
    unsigned long long param_epoch = 0; // bumped when a scalar baked into the kernel nodes changes (fs_set_params)
    long stat_graph_updates = 0, stat_graph_instantiations = 0;
    // Slab frames (world > 1): the exchange-free runs of kernel launches of a plan ("segments") are recorded as hipGraphs
    // and replayed, keyed by everything that decides their launches (ping-pong roles at plan start, the steps' row
    // ranges and sweep counts, the scalar parameters); exchanges, joins and the CFL reduction stay eager between them.
    std::unordered_map<unsigned long long, hipGraphExec_t> seg_graphs;
    unsigned long long plan_salt = 0;   // set by run_frames before a plan is executed: which plan, from which roles
    long stat_seg_replays = 0, stat_seg_captures = 0, stat_seg_eager = 0;
    int seg_min_stable = 2;             // frames the parameters must have stayed put before segments are recorded
    // The speculated advection halo of frame f is checked against the all-reduced CFL bound only at the next call that
    // is not fs_add_force (fs_step of frame f + 1, fs_sync, a read ...): the host does not wait for the GPU inside
    // fs_step.  Until then the roles from before the advection are kept, and the splats applied since are logged, so
    // that a guess that was too small can be repaired: advection planned again with the true depth, splats re-applied.
    int defer_check = 1;                // FS_OPT_SLAB_DEFER
    bool chk_pending = false;
    int chk_h = 0, chk_slot = 0;
    fs::Field chk_saved[fs::F_COUNT];
    std::vector<Splat> splat_log;
    // statistics of the slab path
    long stat_exchanges = 0, stat_rows = 0, stat_frames = 0, stat_redo = 0;
    double host_part_us[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // fs_slab_host_breakdown: planning, replays, recordings, eager runs, exchanges, CFL reduction, joins, tracers
    double host_frame_us = 0.0, host_wait_us = 0.0;   // host time inside run_frames / of that blocked on the CFL value (fs_slab_host_times)
    long host_frames = 0;
    int last_halo = 0;
    // tracers
    long max_particles = 0, n_particles = 0;
    float *part[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    fs_particle_params pparams{};
    // tracers on a slab: partner arrays (the "keep" side of a re-homing), per-destination outboxes, counters
    float *part_t[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    float *outbox[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int *tr_counts_dev = nullptr;       // `world` ints: counts, then cursors
    int *tr_counts_host = nullptr;      // pinned, `world` ints: how many of my tracers go to each rank
    int *tr_matrix_dev = nullptr;       // RCCL: world x world counts, summed over ranks (ncclInt32)
    int *tr_matrix_host = nullptr;      // pinned
    std::vector<long> tr_send_off, tr_recv_cnt;
    long stat_migrated = 0;
    // tracer order (one GPU): the arrays are kept sorted by the cache line their velocity lookup reads (passes.hip)
    int sort_every = 32;                // FS_OPT_TRACER_SORT: frames between two sorts (0 = never)
    int frames_since_sort = 1 << 20;
    bool part_sorted = false;           // part[] is in sorted order; part_perm[i] = slot tracer i came from
    int part_flip = 0;                  // toggles with every part <-> part_t swap (part of the frame graphs' key)
    int *part_perm = nullptr, *part_perm_t = nullptr, *sort_keys = nullptr, *sort_hist = nullptr;
    long sort_cap = 0, sort_hist_cap = 0;
    long stat_sorts = 0;
    // fs_particles_spawn while there is room appends without compacting (no host round trip, the frame graphs stay): the
    // slots that were dead at that moment are marked removed and dropped by the next call that shows the arrays or their
    // count to the host (tracer_settle_removed).  n_particles counts SLOTS (removed ones included) until then.
    bool removed_pending = false;
    int *spawn_sums = nullptr;          // fs_particles_spawn: live counts per block of 256 slots, then their scan (blocks + 1 ints)
    long spawn_sums_cap = 0;
    float *lines_scratch = nullptr;     // this device's copy of the static staging array (fs_particle_lines_static); not owned
    hipEvent_t ev0 = nullptr, ev1 = nullptr;  // fs_timer_*
    char error[512] = "";
};

namespace fs {

int ctx_fail(fs_ctx *c, int rc);
fs_grid grid_of(const fs_ctx *c, int depth);

// tracer order (passes.hip)
int tracer_bins(const fs_grid *g, int *nbx, int *shift);
int tracer_sort(const fs_grid *g, const fs_particle_params *pp, long n, float *const in[7], float *const out[7], int *keys, int *hist,
                const int *perm_in, int *perm_out, hipStream_t st);
int tracer_unsort(long n, float *const in[7], float *const out[7], const int *perm, hipStream_t st);
// tracer spawning (passes.hip)
long tracer_blocks(long n);
int tracer_mark_removed(long n, float *life, hipStream_t st);
int tracer_count_live(long n, const float *life, int *sums, bool removed_only, hipStream_t st);
int tracer_compact(long n, float *const in[7], float *const out[7], const int *sums, long drop, bool removed_only, hipStream_t st);
int tracer_spawn(long first, long j0, long count, float *const out[7], float x, float y, float radius, unsigned seed, int *perm, hipStream_t st);

// frame planner (slab.hip)
void plan_update_a(fs_ctx *c, int h_spec);   // update() up to subtractGradient (+ the CFL reduction on slabs);
                                             // h_spec >= 0: prefetch the dye halo of that depth
void plan_update_b(fs_ctx *c, int h);   // advection .. particles, with advection halo h
int run_frame(fs_ctx *c);               // plan + execute one update() on this context (world 1 or RCCL)
int run_frames(fs_ctx **cs, int n);     // ... on all n slabs of an in-process group, in lock step
int settle(fs_ctx *c);                  // the deferred halo check of the last frame, if one is pending (this rank, or the whole local group)
void drop_segment_graphs(fs_ctx *c);
bool bands_wanted(const fs_ctx *c, int N);
void ensure_band_planes(fs_ctx *c, int N);   // is a projection of N sweeps on this context run as two row bands?

}  // namespace fs

#define CTX(c)                                             \
    do {                                                   \
        if (!(c)) { fs::set_error("ctx is NULL"); return FS_ERR_INVALID; } \
        hipError_t e_ = hipSetDevice((c)->device);         \
        if (e_ != hipSuccess) return fs::ctx_fail(c, fs::hip_fail(e_, "hipSetDevice")); \
    } while (0)
#define TRY(c, call)                        \
    do {                                    \
        int rc_ = (call);                   \
        if (rc_) return fs::ctx_fail(c, rc_);   \
    } while (0)
#define TRYHIP(c, call)                                            \
    do {                                                           \
        hipError_t e_ = (call);                                    \
        if (e_ != hipSuccess) return fs::ctx_fail(c, fs::hip_fail(e_, #call)); \
    } while (0)
