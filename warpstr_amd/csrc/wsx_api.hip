// wsx_api.hip -- C ABI of the MI355X-native WarpSTR caller (include/warpstr_hip.h): handle, HBM
// workspace, chunking, launch sequence.  No compute happens on the host: every stage of the per-read
// pipeline is a HIP kernel (dtw_kernels.hip, mid_kernels.hip); the host only places the states of every automaton
// (wsx_place.h, once per handle), sorts read ids by length (load balance), plans the chunks and sizes the workspace.
//
// Launch sequence per chunk of reads (each chunk on one of the handle's streams, chunks side by side):
//   dtw_fill (unmasked) -> traceback -> mid(pass 1: run statistics, sort, borders, segmentation mask, cost1)
//   -> fit (Givens LSQ cubic) -> eval (rescaled signal) -> dtw_fill (masked, rescaled signal) -> traceback
//   -> mid(pass 2: run statistics, borders, cost2, allele length)
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <numeric>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include <pthread.h>
#include <sched.h>

#include "../../include/warpstr_hip.h"
#include "wsx_device.h"
#include "wsx_place.h"

namespace {

thread_local std::string g_err;

#define HIPCHK(expr)                                                                                              \
    do {                                                                                                          \
        hipError_t e_ = (expr);                                                                                   \
        if (e_ != hipSuccess) {                                                                                   \
            char b_[512];                                                                                         \
            snprintf(b_, sizeof(b_), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);  \
            g_err = b_;                                                                                           \
            return WSX_ERR_HIP;                                                                                   \
        }                                                                                                         \
    } while (0)

static_assert(WSX_STACK_SLOT == WSX_DEV_STACK_SLOT, "placement and kernels agree on the stack slot");

struct Variant { // which DP kernel an automaton uses
    int K = 1, F = 2;
    bool generic = false;
    int FL = 2; // predecessors considered by slots 1..: FL < F when the states with more sit in slot 0 ("split")
    bool pk = false; // packed mask rows (K = 1, F = 2, the states with two predecessors in lanes 0..7): 9 bytes per row
    int lm = 0;      // lane-major placement (wsx_place.h): 1 = slots 0 and K-1 export through LDS, 3 = 0, 1 and K-1, 2 = every slot,
                     // 4 = every slot and two pieces to a lane (stacked)
    // back-pointer scratch of ONE read of T samples, in 64-bit words (even: a read's rows start 16-byte aligned):
    //   register-resident fill: per row F + (K-1)*FL 64-bit wave masks, one spare row (dtw_kernels.hip);
    //   packed rows: 18 words per 16 rows; generic fill: 4 bits per row and state, 8 rows per 32-bit word
    size_t bp_read_words(size_t T) const
    {
        if (generic) return (T / 8 + 1) * (size_t)(K * 32);
        if (pk) return (T / 16 + 1) * 18;
        return (((T + 1) * (size_t)(F + (K - 1) * FL)) + 1) & ~(size_t)1;
    }
    // upper bound, in 32-bit words, for a chunk of `samples` samples in `reads` reads that all took this variant
    size_t bp_words(size_t samples, size_t reads) const
    {
        if (generic) return (samples / 8 + reads + 2) * (size_t)(K * 64);
        if (pk) return (samples / 16 + reads + 4) * 18 * 2;
        return (samples + reads + 64) * (size_t)(F + (K - 1) * FL) * 2 + 2 * reads;
    }
    bool same(const Variant &o) const { return K == o.K && F == o.F && generic == o.generic && FL == o.FL && pk == o.pk && lm == o.lm; }
};

thread_local bool g_alloc_oom = false; // a DeviceBuf of this thread's call ran out of device memory

struct DeviceBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            (void)hipGetLastError(); // (the runtime remembers a failed call until it is asked: the next launch's check would report it)
            want = bytes;
            e = hipMalloc(&p, want);
        }
        if (e == hipSuccess) cap = want;
        else if (e == hipErrorOutOfMemory) {
            g_alloc_oom = true; // (run_batch_retry: a smaller workspace limit and another plan instead of a failed call)
            (void)hipGetLastError();
        }
        return e;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

// Host <-> device transfers of caller-owned (pageable) buffers go through a ring of pinned pieces: a copy straight
// from/to pageable memory blocks the calling thread until the stream reaches it, which serialises the chunks of a batch.
// Uploads: the piece is filled by a few host threads, then sent with an asynchronous copy.  Downloads: the asynchronous
// copy lands in the piece; the piece is copied out to the caller's buffer when it is needed again or at the end.
struct PinRing {
    static constexpr size_t kPiece = 32u << 20;
    struct Slot {
        void *pin = nullptr;
        hipEvent_t ev = nullptr;
        void *user_dst = nullptr; // download: where the piece goes once `ev` has passed
        size_t len = 0;
        bool busy = false;
    };
    std::vector<Slot> slots;
    size_t next = 0;
    int copy_threads = 4;

    hipError_t init(int n_slots)
    {
        if (!slots.empty()) return hipSuccess;
        slots.resize(n_slots);
        for (auto &sl : slots) {
            hipError_t e = hipHostMalloc(&sl.pin, kPiece, hipHostMallocDefault);
            if (e != hipSuccess) return e;
            e = hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming);
            if (e != hipSuccess) return e;
        }
        if (const char *e = getenv("WSX_COPY_THREADS")) copy_threads = std::max(1, atoi(e));
        copy_threads = std::min<int>(copy_threads, std::max(1u, std::thread::hardware_concurrency()));
        return hipSuccess;
    }
    void release()
    {
        for (auto &sl : slots) {
            if (sl.ev) (void)hipEventDestroy(sl.ev);
            if (sl.pin) (void)hipHostFree(sl.pin);
        }
        slots.clear();
    }
    void host_copy(void *dst, const void *src, size_t len) const
    {
        const int nt = len >= (4u << 20) ? copy_threads : 1;
        if (nt <= 1) {
            memcpy(dst, src, len);
            return;
        }
        std::vector<std::thread> th;
        const size_t part = (len / nt + 4095) & ~(size_t)4095;
        for (int t = 1; t < nt; t++) {
            const size_t o = std::min(len, part * t), e = std::min(len, part * (t + 1));
            if (e > o) th.emplace_back([=] { memcpy((char *)dst + o, (const char *)src + o, e - o); });
        }
        memcpy(dst, src, std::min(len, part));
        for (auto &t : th) t.join();
    }
    hipError_t settle(Slot &sl)
    {
        if (!sl.busy) return hipSuccess;
        hipError_t e = hipEventSynchronize(sl.ev);
        if (e != hipSuccess) return e;
        if (sl.user_dst) host_copy(sl.user_dst, sl.pin, sl.len);
        sl.busy = false;
        sl.user_dst = nullptr;
        return hipSuccess;
    }
    hipError_t upload(void *dev_dst, const void *host_src, size_t bytes, hipStream_t s)
    {
        for (size_t o = 0; o < bytes; o += kPiece) {
            Slot &sl = slots[next++ % slots.size()];
            hipError_t e = settle(sl);
            if (e != hipSuccess) return e;
            const size_t len = std::min(kPiece, bytes - o);
            host_copy(sl.pin, (const char *)host_src + o, len);
            if ((e = hipMemcpyAsync((char *)dev_dst + o, sl.pin, len, hipMemcpyHostToDevice, s)) != hipSuccess) return e;
            if ((e = hipEventRecord(sl.ev, s)) != hipSuccess) return e;
            sl.busy = true;
        }
        return hipSuccess;
    }
    // The same from reads that live in separate host arrays (reads first .. first+count-1 of `ptrs`, laid out back to back
    // by `offsets`): the copy threads gather straight into the pinned piece, no intermediate buffer on the caller's side.
    hipError_t upload_gather(void *dev_dst, const double *const *ptrs, const int64_t *offsets, int64_t first, int64_t count,
                             hipStream_t s)
    {
        const int64_t s0 = offsets[first], s1 = offsets[first + count]; // samples
        const size_t bytes = (size_t)(s1 - s0) * 8;
        for (size_t o = 0; o < bytes; o += kPiece) {
            Slot &sl = slots[next++ % slots.size()];
            hipError_t e = settle(sl);
            if (e != hipSuccess) return e;
            const size_t len = std::min(kPiece, bytes - o);
            const int64_t a = s0 + (int64_t)(o / 8), b = a + (int64_t)(len / 8); // global sample range of this piece
            auto part = [&](int64_t pa, int64_t pb) { // samples [pa, pb) into the piece
                int64_t r = std::upper_bound(offsets + first, offsets + first + count + 1, pa) - offsets - 1;
                while (pa < pb) {
                    const int64_t e2 = std::min(pb, offsets[r + 1]);
                    if (e2 > pa) memcpy((char *)sl.pin + (size_t)(pa - a) * 8, ptrs[r] + (pa - offsets[r]), (size_t)(e2 - pa) * 8);
                    pa = e2;
                    r++;
                }
            };
            const int nt = len >= (4u << 20) ? copy_threads : 1;
            if (nt <= 1) {
                part(a, b);
            } else {
                std::vector<std::thread> th;
                const int64_t step = (b - a + nt - 1) / nt;
                for (int t = 1; t < nt; t++) {
                    const int64_t pa = std::min(b, a + step * t), pb = std::min(b, a + step * (t + 1));
                    if (pb > pa) th.emplace_back([=] { part(pa, pb); });
                }
                part(a, std::min(b, a + step));
                for (auto &t : th) t.join();
            }
            if ((e = hipMemcpyAsync((char *)dev_dst + o, sl.pin, len, hipMemcpyHostToDevice, s)) != hipSuccess) return e;
            if ((e = hipEventRecord(sl.ev, s)) != hipSuccess) return e;
            sl.busy = true;
        }
        return hipSuccess;
    }
    hipError_t download(void *host_dst, const void *dev_src, size_t bytes, hipStream_t s)
    {
        for (size_t o = 0; o < bytes; o += kPiece) {
            Slot &sl = slots[next++ % slots.size()];
            hipError_t e = settle(sl);
            if (e != hipSuccess) return e;
            const size_t len = std::min(kPiece, bytes - o);
            if ((e = hipMemcpyAsync(sl.pin, (const char *)dev_src + o, len, hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
            if ((e = hipEventRecord(sl.ev, s)) != hipSuccess) return e;
            sl.busy = true;
            sl.user_dst = (char *)host_dst + o;
            sl.len = len;
        }
        return hipSuccess;
    }
    void discard() // pieces left over by a call that failed half-way: their destinations are gone
    {
        for (auto &sl : slots) {
            if (sl.busy) (void)hipEventSynchronize(sl.ev);
            sl.busy = false;
            sl.user_dst = nullptr;
        }
    }
    hipError_t drain()
    {
        for (size_t k = 0; k < slots.size(); k++) { // oldest first
            hipError_t e = settle(slots[(next + k) % slots.size()]);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
};

} // namespace

struct wsx_caller {
    int device = 0;
    hipStream_t stream = nullptr;
    wsx_params prm{};
    std::vector<DevAutomaton> host_aut; // device pointers inside
    std::vector<Variant> variant;
    std::vector<int> n_states;
    std::vector<Variant> uvar; // the distinct kernel variants among `variant` (each has a back-pointer region of its own)
    WsxTuning tun;             // launch-policy knobs (wsx_caller_set_tuning)
    double create_s[5] = {0, 0, 0, 0, 0}; // wsx_caller_create_times
    DeviceBuf zstd_status;              // wsx_zstd_decode: see wsx_internal_zstd_status
    DeviceBuf aut_blob, aut_table;      // the most recent blob / the table of all automata so far
    std::vector<DeviceBuf> aut_retired; // blobs of earlier wsx_caller_add_automata calls (referenced by the table for good) and tables
                                        // that calls still in flight may read: released with the handle
    uint64_t ws_limit = 16ull << 30; // set from the device's free memory at creation (wsx_caller_set_workspace_limit overrides)
    // workspace
    // Up to WSX_MAX_STREAMS workspace sets: consecutive chunks rotate over the handle's stream and internal ones,
    // so that the latency/bandwidth-bound stages of some chunks (traceback, run statistics, fit, ...) run under the
    // VALU-bound DP fill of others.
    struct Work {
        DeviceBuf samples, reads, bp, stage_sig, stage_out, reps;
        DeviceBuf smooth; // fit_smooth_kernel's arrays (rescaling.threshold > 1, sized when a chunk has such reads)
    } work[WSX_MAX_STREAMS];
    int32_t *smooth_host = nullptr; // pinned: per work set, the chunk's {reads for fit_smooth_kernel, most points among them}
    hipStream_t aux[WSX_MAX_STREAMS] = {};  // aux[0] unused (the handle's stream)
    hipEvent_t ev_joins[WSX_MAX_STREAMS] = {};
    int n_streams = 4;        // streams / work sets the handle may use
    int streams_per_call = 4; // chunks of one call that run side by side
    // offsets / automaton ids / launch order of a call: device copy + pinned staging (caller buffers are not kept).
    // Two slots: pipelined calls alternate, so that call k+1 is prepared and enqueued while call k still runs.
    static constexpr int kMetaSlots = 4;
    static constexpr int64_t kSmallPipeSamples = (int64_t)52 << 20; // pipelined calls up to this size: one chunk, four in flight
    DeviceBuf meta[kMetaSlots];
    void *pinned[kMetaSlots] = {};
    size_t pinned_cap[kMetaSlots] = {};
    hipEvent_t ev_meta[kMetaSlots] = {}; // recorded when the call that used the slot has finished
    int in_flight = 2;                    // pipelined calls the host may run ahead of the device (<= kMetaSlots)
    int in_flight_small = kMetaSlots;     // ... for small batches (one chunk each)
    int64_t small_pipe_samples = kSmallPipeSamples; // where the one-chunk policy for small pipelined calls ends (WSX_TUNE_SMALL_PIPE_SAMPLES)
    bool pipelined = false;                     // wsx_caller_set_pipelined
    uint64_t call_seq = 0;
    int rot = 0; // pipelined calls with fewer chunks than streams: the first work set / stream of the next call
    int chunks_override = 0; // WSX_CHUNKS at creation: chunks per call (tuning knob)
    hipStream_t join_st = nullptr; // pipelined calls end here instead of on the handle's stream
    hipEvent_t ev_fork = nullptr;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> dp_events;
    std::vector<int32_t> dp_reads; // reads of the fill launch between the pair
    size_t dp_events_used = 0;
    bool timing_valid = false;
    int64_t last_samples = 0; // samples of the most recent call (wsx_caller_workspace)
    bool timing_window = false; // wsx_caller_timing_window: the fill events of successive calls accumulate
    hipEvent_t ev_window = nullptr; // start of the window (the first call's begin)
    bool window_open = false;
    int max_states = 0;
    bool have_bases = true; // every automaton came with last_base
    std::vector<hipEvent_t> sched_events;
    size_t sched_used = 0;
    PinRing ring_up, ring_down; // host-buffer calls only (allocated on first use)
    DeviceBuf prep_pool[12];    // wsx_prepare_signals: histograms and per-chunk buffers, kept between calls
    // ... and its metadata staging (pinned): a ring, each slot with the event recorded after its last use -- with one buffer a call
    // waited for the call before it to have uploaded its metadata, i.e. for most of the batch before it (a driver that submits
    // batch after batch lost 2-10 ms a batch there)
    static constexpr int kPrepSlots = 4;
    void *prep_pinned[kPrepSlots] = {};
    size_t prep_pinned_cap[kPrepSlots] = {};
    hipEvent_t ev_prep[kPrepSlots] = {};
    unsigned prep_turn = 0;
    struct VbzSlot {             // wsx_vbz_decode: block tables on their way to the device (three calls may be in flight)
        void *host = nullptr;
        size_t host_cap = 0;
        DeviceBuf dev;
        hipEvent_t ev = nullptr;
    } vbz_ring[8];   // (two calls a batch -- wsx_zstd_decode, wsx_vbz_decode --, three batches in flight)
    unsigned vbz_turn = 0;
    void *pinned_res = nullptr; // host-buffer calls: the batch's result records land here first
    size_t pinned_res_cap = 0;
};

namespace {

int set_device(wsx_caller *c) { HIPCHK(hipSetDevice(c->device)); return WSX_SUCCESS; }

size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// carve typed arrays out of one allocation
struct Carver {
    char *base;
    size_t used = 0;
    explicit Carver(void *p) : base((char *)p) {}
    template <class T> T *take(size_t n)
    {
        T *p = (T *)(base + used);
        used += align_up(n * sizeof(T));
        return p;
    }
};

struct ChunkPlan {
    int64_t first, count;     // reads
    int64_t base_off, samples; // samples
    int max_T;
};

int get_event_pair(wsx_caller *c, hipEvent_t *a, hipEvent_t *b, int32_t reads)
{
    if (c->dp_events_used == c->dp_events.size()) {
        hipEvent_t x, y;
        HIPCHK(hipEventCreate(&x));
        HIPCHK(hipEventCreate(&y));
        c->dp_events.push_back({x, y});
        c->dp_reads.push_back(0);
    }
    *a = c->dp_events[c->dp_events_used].first;
    *b = c->dp_events[c->dp_events_used].second;
    c->dp_reads[c->dp_events_used] = reads;
    c->dp_events_used++;
    return WSX_SUCCESS;
}

// bytes of workspace per sample of ONE work set, as run_batch allocates it: the per-sample arrays, the back-pointer scratch
// of every distinct kernel variant of the handle (each has a region of its own), and for host buffers the staging areas
size_t per_sample_bytes(const wsx_caller *c, bool host_mem, bool want_traces)
{
    size_t bp = 0; // (the reads of a chunk lie back to back in ONE region whatever variant each takes: the widest decides)
    for (auto &v : c->uvar) bp = std::max(bp, v.bp_words(4096, 0) * 4 / 4096 + 1);
    size_t b = 8 /*rescaled*/ + 2 + 4 /*runs*/ + 24 + 1 /*alignment*/ + 16 /*fit pairs*/ + 16 /*scratch*/ +
               1 /*mask bits, rounded up*/ + bp;
    if (host_mem) b += 8 /*signal staging*/ + (want_traces ? (2 + 2 + 8 + 3) : 0);
    return b + 8; // alignment slack
}
// ... and per read: the per-read arrays, the spare rows of the back-pointer regions, reps_as_one scratch, staged rows
size_t per_read_bytes(const wsx_caller *c, bool host_mem, size_t last_row_bytes)
{
    size_t b = 168 + sizeof(wsx_result) + 64;
    size_t spare = 0;
    for (auto &v : c->uvar) spare = std::max(spare, v.bp_words(0, 1) * 4);
    b += spare;
    if (c->prm.reps_as_one) b += 2 * (size_t)c->max_states * sizeof(int32_t);
    if (host_mem) b += 16 + last_row_bytes;
    return b;
}

// A new thread starts on its parent's CPU and waits for the kernel's load balancer to move it, which on virtualised hosts takes
// longer than the placement of a handle lasts (16 threads then share one CPU).  Move thread t to the t-th CPU of the process's
// affinity mask and restore the mask at once: nothing stays pinned.
void spread_thread(int t)
{
    cpu_set_t all;
    CPU_ZERO(&all);
    if (sched_getaffinity(0, sizeof all, &all) != 0) return;
    const int n = CPU_COUNT(&all);
    if (n < 2) return;
    int local_rank = 0;
    if (const char *e = getenv("LOCAL_RANK")) local_rank = atoi(e);
    int want = (t + 1 + 16 * local_rank) % n, cpu = -1;
    for (int c = 0; c < CPU_SETSIZE; c++)
        if (CPU_ISSET(c, &all) && want-- == 0) { cpu = c; break; }
    if (cpu < 0) return;
    cpu_set_t one;
    CPU_ZERO(&one);
    CPU_SET(cpu, &one);
    if (pthread_setaffinity_np(pthread_self(), sizeof one, &one) == 0) (void)pthread_setaffinity_np(pthread_self(), sizeof all, &all);
}

__global__ void pack_mask_kernel(const uint8_t *mask, const int64_t *offsets, int first_read, int64_t base_off, int n,
                                 uint32_t *bits)
{
    const int lr = blockIdx.x;
    if (lr >= n) return;
    const int r = first_read + lr;
    const long long off = offsets[r] - base_off;
    const int T = (int)(offsets[r + 1] - offsets[r]);
    const int w = blockIdx.y * blockDim.x + threadIdx.x;
    if (w * 32 >= T) return;
    uint32_t v = 0;
    for (int b = 0; b < 32 && w * 32 + b < T; b++) v |= (mask[off + w * 32 + b] ? 1u : 0u) << b;
    bits[off / 32 + lr + w] = v;
}

// Automata appended to a handle: validated, placed (wsx_place.h), packed into a device blob of their own, the table of all
// automata so far uploaded anew.  wsx_caller_create's first part, and wsx_caller_add_automata.
static int append_automata(wsx_caller *c, const wsx_automaton *automata, int32_t n_automata)
{
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(now() - t).count(); };
    auto t_phase = now();
    // validate + size the automaton blob
    size_t blob = 0;
    for (int a = 0; a < n_automata; a++) {
        const wsx_automaton &A = automata[a];
        if (A.n_states <= 0 || A.n_states > 65535 || A.endstate < 0 || A.endstate >= A.n_states || !A.value ||
            !A.seq_idx || !A.pred_ptr || !A.pred_idx || !A.repeat_mask) {
            g_err = "wsx_caller_create / wsx_caller_add_automata: malformed automaton";
            return WSX_ERR_INVALID;
        }
        const int S = A.n_states;
        if (A.pred_ptr[0] != 0) {
            g_err = "wsx_caller_create / wsx_caller_add_automata: pred_ptr[0] must be 0";
            return WSX_ERR_INVALID;
        }
        for (int j = 0; j < S; j++)
            if (A.pred_ptr[j + 1] < A.pred_ptr[j]) {
                g_err = "wsx_caller_create / wsx_caller_add_automata: pred_ptr not monotone";
                return WSX_ERR_INVALID;
            }
        const int E = A.pred_ptr[S]; // (only now known to be >= 0)
        for (int e = 0; e < E; e++)
            if (A.pred_idx[e] < 0 || A.pred_idx[e] >= S) {
                g_err = "wsx_caller_create / wsx_caller_add_automata: predecessor index out of range";
                return WSX_ERR_INVALID;
            }
        blob += align_up(S * 8) + align_up(S * 4) + align_up((S + 1) * 4) + align_up(std::max(E, 1) * 4) + align_up(S) +
                align_up((size_t)((S + 63) / 64) * 64 * 8) + align_up(S) + 2 * align_up(((S + 63) / 64) * 64 * 2 + 2 * S) +
                align_up((size_t)((S + 63) / 64) * WSX_MAX_F * 64 * 2) + align_up((size_t)((S + 63) / 64) * 64 * 2);
    }
    // a blob of its own per call (the table's entries point into it for the life of the handle) and a NEW table for all automata
    // so far: calls still in flight read the old one
    const size_t base = c->host_aut.size();   // index of the first automaton of this call
    DeviceBuf new_blob, new_table;
    struct Undo {   // a failure below leaves the handle as it was
        wsx_caller *c;
        size_t n_aut, n_uvar;
        int max_states;
        bool have_bases;
        DeviceBuf *blob, *table;
        bool armed = true;
        ~Undo()
        {
            if (!armed) return;
            c->host_aut.resize(n_aut);
            c->variant.resize(n_aut);
            c->n_states.resize(n_aut);
            c->uvar.resize(n_uvar);
            c->max_states = max_states;
            c->have_bases = have_bases;
            blob->release();
            table->release();
        }
    } undo{c, base, c->uvar.size(), c->max_states, c->have_bases, &new_blob, &new_table};
    HIPCHK(new_blob.ensure(blob));
    HIPCHK(new_table.ensure(sizeof(DevAutomaton) * (base + (size_t)n_automata)));
    std::unique_ptr<char[]> hblob_mem(new char[blob]); // (not value-initialised: every byte the device reads is written below)
    char *const hblob = hblob_mem.get();
    c->create_s[0] += since(t_phase);
    t_phase = now();
    size_t used = 0;
    auto put = [&](const void *src, size_t bytes) -> void * {
        void *d = (char *)new_blob.p + used;
        memcpy(hblob + used, src, bytes);
        used += align_up(bytes);
        return d;
    };
    // Where the states of every automaton live (wsx_place.h) is host work of a millisecond or two per automaton and
    // independent between automata: a handle for all loci of a run (thousands of automata, warpstr_amd/loci.py) places them
    // on several host threads.
    struct Placed {
        Variant v;
        int mf = 0;
        std::vector<uint16_t> pos, wslot, state_at, paddr;
        std::vector<uint64_t> p4;
        bool store_pos = false;
        uint64_t stack_mask = 0;
    };
    std::vector<Placed> placed((size_t)n_automata);
    auto place_one = [&](int a) {
        const wsx_automaton &A = automata[a];
        const int S = A.n_states;
        Placed &P = placed[a];
        int mf = 0;
        for (int j = 0; j < S; j++) mf = std::max(mf, A.pred_ptr[j + 1] - A.pred_ptr[j]);
        P.mf = mf;
        Variant &v = P.v;
        v.K = (S + 63) / 64;
        v.F = std::max(mf, 1);
        v.generic = !wsx_fast_pass_supported(c->prm.min_values_per_state, v.K, v.F);
        // Several slots and only a few states with many predecessors (loop entries, IUPAC alternatives): give those
        // states slot 0 (positions 0..63), so that only that slot pays for the extra candidates.  Slots 1.. then
        // consider FL predecessors: 1 when every state with two or more fits in slot 0, else 2.
        const int Fk = v.F <= 2 ? 2 : v.F;
        v.FL = Fk;
        if (!v.generic && wsx_split_supported(c->prm.min_values_per_state, v.K) && !wsx_exp_env("WSX_NO_SPLIT")) {
            int n_ge2 = 0, n_gt2 = 0;
            for (int j = 0; j < S; j++) {
                const int nf = A.pred_ptr[j + 1] - A.pred_ptr[j];
                n_ge2 += nf >= 2;
                n_gt2 += nf > 2;
            }
            if (n_ge2 <= 64) v.FL = 1;
            else if (Fk > 2 && n_gt2 <= 64) v.FL = 2;
        }
        // where the states live: position (slot, lane) and LDS export slot of every state (wsx_place.h)
        std::vector<uint16_t> &pos = P.pos, &wslot = P.wslot;
        pos.resize(S);
        wslot.resize((size_t)v.K * 64);
        std::iota(pos.begin(), pos.end(), (uint16_t)0);
        std::iota(wslot.begin(), wslot.end(), (uint16_t)0);
        if (!v.generic) {
            const bool want_pk = v.K == 1 && Fk == 2 && !wsx_exp_env("WSX_NO_PACK");
            // several slots: the lane-major layout (chains along the slots of a lane, LDS only for slot 0) where it fits
            WsxLanePlacement lp;
            const int lm_mode = [] {
                const char *e = wsx_exp_env("WSX_FILL_LM");
                return e ? atoi(e) : 1; // 0: off; 1: on; 2: on, every slot exports
            }();
            if (lm_mode != 0 && wsx_lane_major_supported(c->prm.min_values_per_state, v.K))
                lp = wsx_place_lane_major(S, A.pred_ptr, A.pred_idx, v.K);
            // ... and for five-slot automata with many short chains and little room (DM2 at flank 110: 266 states) its
            // stacked form: two pieces to a lane, the upper one's first state reading LDS in slot 2, every slot exporting.
            // 20 000 reads x 3 000 samples, same box (profiles/r03_stacked_ab.log): DM2 17.0 -> 16.5 ms per call; with four
            // slots it LOSES to the slot-major kernel (HD, both strands: 12.1 vs 11.8 ms -- four exports and the extra read
            // load the LDS pipe as much as the slot-major exchange does), so four-slot automata keep that one.
            const int stacked_min_k = [] {
                const char *e = wsx_exp_env("WSX_STACKED_MIN_K");
                return e ? atoi(e) : 5;
            }();
            if (lp.lm == 0 && lm_mode != 0 && wsx_lane_major_supported(c->prm.min_values_per_state, v.K) && v.K >= stacked_min_k &&
                !wsx_exp_env("WSX_NO_STACKED"))
                lp = wsx_place_lane_stacked(S, A.pred_ptr, A.pred_idx, v.K);
            if (lp.lm != 0) {
                v.FL = 1;
                v.lm = lp.lm == 4 ? 4 : lm_mode == 2 ? 2 : (lp.lm == 3 && v.K < 4 ? 2 : lp.lm);
                P.stack_mask = lp.stack_mask;
            }
            const WsxPlacement pl = lp.lm != 0 ? lp.pl
                                    : wsx_exp_env("WSX_PLAIN_PLACEMENT") && v.FL >= Fk
                                        ? WsxPlacement{}
                                        : wsx_place_states(S, A.pred_ptr, A.pred_idx, v.K, Fk, v.FL, want_pk);
            if (!pl.pos.empty()) {
                pos = pl.pos;
                wslot = pl.wslot;
                v.pk = want_pk && pl.low8;
                if (!pl.identity) {
                    P.store_pos = true;
                    P.state_at = pl.state_at;
                }
            }
        }
        // pred4: the walk of the mask traceback runs in position space -- per position (slot*64 + lane) the positions of
        // its state's first four predecessors, 16 bits each
        P.p4.assign((size_t)v.K * 64, 0);
        for (int j = 0; j < S; j++)
            for (int e = A.pred_ptr[j], t = 0; e < A.pred_ptr[j + 1] && t < 4; e++, t++)
                P.p4[pos[j]] |= (uint64_t)pos[A.pred_idx[e]] << (16 * t);
        // paddr: which LDS export slot (slot k, predecessor f, lane) reads in the register-resident fill.  A ds_read_b64
        // serves lanes 0-31 and 32-63 in one cycle each when no two lanes of a group hit the same bank pair (slot mod 32)
        // at different addresses; lanes without predecessor f all read the same +inf slot (a broadcast), picked among the
        // 32 spare slots K*64.. so that its bank pair is one the group's real readers leave free.
        if (!v.generic) {
            std::vector<uint16_t> &paddr = P.paddr;
            paddr.assign((size_t)v.K * WSX_MAX_F * 64, (uint16_t)(v.K * 64));
            std::vector<int> state_of((size_t)v.K * 64, -1);
            for (int j = 0; j < S; j++) state_of[pos[j]] = j;
            for (int k = 0; k < v.K; k++)
                for (int f = 0; f < WSX_MAX_F; f++)
                    for (int g = 0; g < 2; g++) {
                        int used[32] = {0};
                        // (lane-major layouts: a state above slot 0 takes its predecessor from a register; in the stacked
                        // layout the stack slot's ds_read_b64 is issued by EVERY lane, and the lanes that do not start a piece
                        // there read the +inf slot -- their in-lane predecessor's export slot would sit on the bank pair of
                        // their own lane and collide with what the piece heads read: 2 conflict cycles per row on DM2)
                        auto reads_lds = [&](int l, int j) {
                            if (j < 0 || A.pred_ptr[j + 1] - A.pred_ptr[j] <= f) return false;
                            if (v.lm != 0 && k > 0) return v.lm == 4 && k == WSX_STACK_SLOT && ((P.stack_mask >> l) & 1ull) != 0;
                            return true;
                        };
                        for (int l = g * 32; l < g * 32 + 32; l++) {
                            const int j = state_of[k * 64 + l];
                            if (!reads_lds(l, j)) continue;
                            const int pa = wslot[pos[A.pred_idx[A.pred_ptr[j] + f]]];
                            paddr[((size_t)k * WSX_MAX_F + f) * 64 + l] = (uint16_t)pa;
                            used[pa & 31]++;
                        }
                        int best = 0;
                        for (int r = 1; r < 32; r++)
                            if (used[r] < used[best]) best = r;
                        for (int l = g * 32; l < g * 32 + 32; l++) {
                            const int j = state_of[k * 64 + l];
                            if (!reads_lds(l, j)) paddr[((size_t)k * WSX_MAX_F + f) * 64 + l] = (uint16_t)(v.K * 64 + best);
                        }
                    }
        }
    };
    // Where a state lives depends on the GRAPH of its automaton alone (the predecessor lists), not on the levels: the loci of a
    // panel repeat a few dozen patterns, and the flanks of a locus change its levels far more often than its graph.  Automata
    // with the same graph are placed once (found by a hash of the lists, confirmed by comparing them) -- and what follows from
    // the graph alone (the lists themselves, positions, predecessor words, LDS addresses) is in the device blob once.
    std::vector<int> rep((size_t)n_automata);
    {
        std::vector<int> todo;
        {
            std::unordered_map<uint64_t, std::vector<int>> seen;
            for (int a = 0; a < n_automata; a++) {
                const wsx_automaton &A = automata[a];
                const int S = A.n_states, E = A.pred_ptr[S];
                uint64_t h = 1469598103934665603ull ^ (uint64_t)S;
                for (int j = 0; j <= S; j++) h = (h ^ (uint64_t)(uint32_t)A.pred_ptr[j]) * 1099511628211ull;
                for (int e = 0; e < E; e++) h = (h ^ (uint64_t)(uint32_t)A.pred_idx[e]) * 1099511628211ull;
                rep[a] = a;
                for (int b : seen[h]) {
                    const wsx_automaton &B = automata[b];
                    if (B.n_states == S && memcmp(B.pred_ptr, A.pred_ptr, (size_t)(S + 1) * 4) == 0 &&
                        memcmp(B.pred_idx, A.pred_idx, (size_t)E * 4) == 0) {
                        rep[a] = b;
                        break;
                    }
                }
                if (rep[a] == a) {
                    seen[h].push_back(a);
                    todo.push_back(a);
                }
            }
        }
        const int n_todo = (int)todo.size();
        const int nt = n_todo >= 16 ? (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u) : 1;
        if (nt <= 1) {
            for (int a : todo) place_one(a);
        } else {
            // (taken one at a time from a shared counter: a five-slot automaton with nested loops takes a hundred times as long
            // as a four-slot chain, and the loci of a run repeat with a period that a fixed stride maps onto a few threads)
            std::vector<std::thread> th;
            std::vector<std::exception_ptr> errs((size_t)nt);
            std::atomic<int> next_a{0};
            for (int t = 0; t < nt; t++)
                th.emplace_back([&, t] {
                    try {
                        spread_thread(t);
                        for (int q = next_a.fetch_add(1); q < n_todo; q = next_a.fetch_add(1)) place_one(todo[q]);
                    } catch (...) {
                        errs[t] = std::current_exception();
                    }
                });
            for (auto &t : th) t.join();
            for (auto &e : errs)
                if (e) std::rethrow_exception(e);
        }
    }
    c->create_s[1] += since(t_phase);
    t_phase = now();
    c->host_aut.reserve(base + (size_t)n_automata);
    for (int a = 0; a < n_automata; a++) {
        const wsx_automaton &A = automata[a];
        const int S = A.n_states, E = A.pred_ptr[S];
        const bool first = rep[a] == a;   // (a representative comes before the automata it stands for)
        const Placed &P = placed[rep[a]];
        Variant v = P.v;
        const int mf = P.mf;
        DevAutomaton D{};
        D.n_states = S;
        D.endstate = A.endstate;
        D.flank_length = A.flank_length;
        D.max_fanin = mf;
        D.seq_idx_last = A.seq_idx[S - 1];
        D.reverse = A.reverse ? 1 : 0;
        D.value = (const double *)put(A.value, (size_t)S * 8);
        D.seq_idx = (const int32_t *)put(A.seq_idx, (size_t)S * 4);
        if (!first) {
            const DevAutomaton &R = c->host_aut[base + (size_t)rep[a]];
            D.pred_ptr = R.pred_ptr, D.pred_idx = R.pred_idx;
            D.pos = R.pos, D.state_at = R.state_at, D.wslot = R.wslot, D.pred4 = R.pred4, D.paddr = R.paddr;
        } else {
            D.pred_ptr = (const int32_t *)put(A.pred_ptr, (size_t)(S + 1) * 4);
            int32_t dummy = 0;
            D.pred_idx = (const int32_t *)put(E ? (const void *)A.pred_idx : (const void *)&dummy, (size_t)std::max(E, 1) * 4);
        }
        D.repeat_mask = (const uint8_t *)put(A.repeat_mask, (size_t)S);
        D.last_base = A.last_base ? (const uint8_t *)put(A.last_base, (size_t)S) : nullptr;
        if (!A.last_base) c->have_bases = false;
        D.stack_mask = P.stack_mask;
        if (first) {
            D.pos = nullptr;
            D.state_at = nullptr;
            if (P.store_pos) {
                D.pos = (const uint16_t *)put(P.pos.data(), (size_t)S * 2);
                D.state_at = (const uint16_t *)put(P.state_at.data(), P.state_at.size() * 2);
            }
            if (!v.generic && v.K == 1) D.wslot = (const uint16_t *)put(P.wslot.data(), P.wslot.size() * 2);
            D.pred4 = (const uint64_t *)put(P.p4.data(), P.p4.size() * 8);
            if (!v.generic) D.paddr = (const uint16_t *)put(P.paddr.data(), P.paddr.size() * 2);
        }
        c->host_aut.push_back(D);
        if (v.generic) {
            if (mf > 15) {
                g_err = "automaton fan-in > 15 is not supported";
                return WSX_ERR_UNSUPPORTED;
            }
            if ((size_t)(c->prm.min_values_per_state + 1) * v.K * 64 * 8 + (size_t)v.K * 64 * 4 > 150 * 1024) {
                g_err = "automaton too large for the generic DP kernel's LDS ring";
                return WSX_ERR_UNSUPPORTED;
            }
        } else {
            v.F = v.F <= 2 ? 2 : v.F;
        }
        c->variant.push_back(v);
        c->n_states.push_back(S);
        c->max_states = std::max(c->max_states, S);
    }
    // Fewer launch groups per chunk: every kernel variant among a chunk's reads is a launch of its own on the chunk's stream, and
    // a minority variant's few hundred wavefronts last as long as their longest read (configs[4]: the three- and four-candidate
    // variants, 12 % of the reads, took 1.8 of a pass's 4.3 ms).  Variants that differ only in the number of candidates of slot
    // 0 share the larger kernel where the difference is one candidate (three vector instructions per row for those reads).
    // (Their launches on side streams instead: slower, 17.1-22 vs 16.4 ms per step -- more streams than the runtime's queues.)
    // (the variants of a handle are few and its automata many: the partners are looked for among the distinct variants, as they
    // were before any merge -- with every automaton compared against every other, a handle of 48 000 automata took 2.2 s here)
    if (!wsx_exp_env("WSX_NO_VARIANT_MERGE")) {
        std::vector<Variant> distinct;
        for (const auto &v : c->variant) {
            bool seen = false;
            for (const auto &u : distinct) seen = seen || u.same(v);
            if (!seen) distinct.push_back(v);
        }
        for (size_t q = base; q < c->variant.size(); q++) {   // (the automata of earlier calls keep the kernel their calls in flight use)
            Variant &v = c->variant[q];
            if (v.generic || v.pk || v.K < 2 || v.F < 3) continue;
            const int f0 = v.F;   // (one step: a variant joins the kernel with ONE candidate more, whatever the order of the automata)
            for (const auto &u : distinct)
                if (!u.generic && !u.pk && u.K == v.K && u.FL == v.FL && u.lm == v.lm && u.F == f0 + 1) v.F = u.F;
        }
    }
    for (size_t q = base; q < c->variant.size(); q++) {
        const Variant &v = c->variant[q];
        bool seen = false;
        for (auto &u : c->uvar) seen = seen || u.same(v);
        if (!seen) c->uvar.push_back(v);
    }
    c->create_s[2] += since(t_phase);
    t_phase = now();
    // (synchronous copies from pageable memory: the host waits, the device goes on with whatever it is running)
    HIPCHK(hipMemcpy(new_blob.p, hblob, used, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(new_table.p, c->host_aut.data(), sizeof(DevAutomaton) * c->host_aut.size(), hipMemcpyHostToDevice));
    c->create_s[3] += since(t_phase);
    undo.armed = false;
    if (c->aut_blob.p) c->aut_retired.push_back(c->aut_blob);
    if (c->aut_table.p) c->aut_retired.push_back(c->aut_table);
    c->aut_blob = new_blob;
    c->aut_table = new_table;
    return WSX_SUCCESS;
}

} // namespace

extern "C" {

int wsx_internal_on_exception(void);
int wsx_abi_version(void) { return WSX_ABI_VERSION; }

int wsx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *wsx_last_error(void) { return g_err.c_str(); }

int wsx_caller_create(wsx_caller **out, int device, const wsx_automaton *automata, int32_t n_automata,
                      const wsx_params *params, void *stream)
try {
    if (!out || !automata || n_automata <= 0 || !params) {
        g_err = "wsx_caller_create: null argument";
        return WSX_ERR_INVALID;
    }
    *out = nullptr;
    if (params->min_values_per_state < 2 || params->states_in_segment < 2 || !(params->threshold > 0) ||
        !(params->max_std > 0)) {
        g_err = "wsx_caller_create: invalid parameters (src/config.py:91-119 asserts)";
        return WSX_ERR_INVALID;
    }
    // (rescaling.threshold > 1 is accepted, as upstream accepts it, src/config.py:97-100: a read whose first least-squares fit
    // leaves FITPACK's polynomial branch -- possible only then -- takes fpcurf's knot-adding and smoothing branch in
    // fit_smooth_kernel)
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        g_err = "no HIP device available";
        return WSX_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= ndev) {
        g_err = "wsx_caller_create: device index out of range";
        return WSX_ERR_INVALID;
    }
    wsx_caller *c = new wsx_caller();
    struct Guard { // every failure below releases what has been built so far (wsx_caller_destroy takes partial handles)
        wsx_caller *c;
        ~Guard() { if (c) wsx_caller_destroy(c); }
    } guard{c};
    c->device = device;
    c->stream = (hipStream_t)stream;
    c->prm = *params;
    HIPCHK(hipSetDevice(device));
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(now() - t).count(); };
    auto t_phase = now();

    {
        const int rc_add = append_automata(c, automata, n_automata);
        if (rc_add != WSX_SUCCESS) return rc_add;
    }
    t_phase = now();
    // Workspace limit: what the device can give.  A fixed 16 GiB made a 100 000-read call of 2 kSample reads take eight
    // chunks instead of four (more, smaller chunks lose: the serial per-read stages last as long for 6 000 reads as for
    // 100 000); the handle only ever allocates what a call needs, the limit is an upper bound.
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > 0)
            c->ws_limit = std::max<uint64_t>(2ull << 30, (uint64_t)((double)free_b * 0.6));
    }
    if (c->prm.threshold > 1.0) HIPCHK(hipHostMalloc((void **)&c->smooth_host, 2 * WSX_MAX_STREAMS * sizeof(int32_t), hipHostMallocDefault));
    c->create_s[2] += since(t_phase);
    t_phase = now();
    HIPCHK(hipEventCreate(&c->ev_begin));
    HIPCHK(hipEventCreate(&c->ev_end));
    for (auto &e : c->ev_meta) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_joins[0], hipEventDisableTiming));
    if (const char *e = getenv("WSX_STREAMS")) c->n_streams = std::min(WSX_MAX_STREAMS, std::max(1, atoi(e)));
    if (const char *e = getenv("WSX_CHUNKS")) c->chunks_override = std::max(1, atoi(e));
    // (experiment builds only, -DWSX_EXPERIMENT: the A/B switches of scripts/)
    if (const char *e = wsx_exp_env("WSX_STREAMS_PER_CALL")) c->streams_per_call = std::min(WSX_MAX_STREAMS, std::max(1, atoi(e)));
    if (const char *e = wsx_exp_env("WSX_INFLIGHT")) c->in_flight = std::min((int)wsx_caller::kMetaSlots, std::max(2, atoi(e)));
    if (const char *e = wsx_exp_env("WSX_INFLIGHT_SMALL")) c->in_flight_small = std::min((int)wsx_caller::kMetaSlots, std::max(2, atoi(e)));
    if (const char *e = wsx_exp_env("WSX_SMALL_PIPE_SAMPLES")) c->small_pipe_samples = (int64_t)atoll(e);
    if (const char *e = wsx_exp_env("WSX_STREAM_TRACEBACK_MIN")) c->tun.stream_traceback_min = atoi(e);
    if (const char *e = wsx_exp_env("WSX_BORDERS_WAVE_BELOW")) c->tun.borders_wave_below = atoi(e);
    if (wsx_exp_env("WSX_SEGMENT_TWO_KERNELS")) c->tun.segment_two_kernels = 1;
    if (const char *e = wsx_exp_env("WSX_FILL_BLOCKS_PER_CU")) c->tun.fill_blocks_per_cu = atoi(e);
    // The streams one call spreads over exist from the start; the others (small pipelined calls taking turns) are created
    // when first used: the runtime maps streams onto a few hardware queues in order of creation, streams that merely exist
    // already cost big calls 2 % (profiles/r02_ab_streams.log), and creating these ones late, after work has been queued,
    // cost 6 %.
    for (int w = 1; w < std::min(c->n_streams, c->streams_per_call); w++) {
        HIPCHK(hipStreamCreateWithFlags(&c->aux[w], hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&c->ev_joins[w], hipEventDisableTiming));
    }
    c->create_s[4] = since(t_phase);
    guard.c = nullptr;
    *out = c;
    return WSX_SUCCESS;
} catch (...) {
    return wsx_internal_on_exception();
}

int wsx_caller_add_automata(wsx_caller *c, const wsx_automaton *automata, int32_t n_automata, int32_t *first_index)
try {
    if (!c || !automata || n_automata <= 0) {
        g_err = "wsx_caller_add_automata: null argument";
        return WSX_ERR_INVALID;
    }
    if (c->host_aut.size() + (size_t)n_automata > (size_t)INT32_MAX) {
        g_err = "wsx_caller_add_automata: too many automata";
        return WSX_ERR_INVALID;
    }
    HIPCHK(hipSetDevice(c->device));
    const int32_t first = (int32_t)c->host_aut.size();
    const int rc = append_automata(c, automata, n_automata);
    if (rc == WSX_SUCCESS && first_index) *first_index = first;
    return rc;
} catch (...) {
    return wsx_internal_on_exception();
}

int wsx_caller_create_times(const wsx_caller *c, double *seconds, int32_t capacity)
{
    if (!c || !seconds || capacity < 0) return WSX_ERR_INVALID;
    for (int i = 0; i < std::min<int32_t>(capacity, 5); i++) seconds[i] = c->create_s[i];
    return WSX_SUCCESS;
}

void wsx_caller_destroy(wsx_caller *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (int w = 1; w < WSX_MAX_STREAMS; w++)
        if (c->aux[w]) (void)hipStreamSynchronize(c->aux[w]);
    if (c->join_st) {
        (void)hipStreamSynchronize(c->join_st);
        (void)hipStreamDestroy(c->join_st);
    }
    c->ring_up.release();
    c->ring_down.release();
    if (c->pinned_res) (void)hipHostFree(c->pinned_res);
    if (c->smooth_host) (void)hipHostFree(c->smooth_host);
    for (int k = 0; k < wsx_caller::kPrepSlots; k++) {
        if (c->prep_pinned[k]) (void)hipHostFree(c->prep_pinned[k]);
        if (c->ev_prep[k]) (void)hipEventDestroy(c->ev_prep[k]);
    }
    for (auto &v : c->vbz_ring) {
        if (v.host) (void)hipHostFree(v.host);
        if (v.ev) (void)hipEventDestroy(v.ev);
        v.dev.release();
    }
    for (DeviceBuf *b : {&c->aut_blob, &c->aut_table, &c->zstd_status}) b->release();
    for (auto &b : c->aut_retired) b.release();
    for (auto &b : c->meta) b.release();
    for (auto &b : c->prep_pool) b.release();
    for (auto &w : c->work)
        for (DeviceBuf *b : {&w.samples, &w.reads, &w.bp, &w.stage_sig, &w.stage_out, &w.reps, &w.smooth}) b->release();
    for (int w = 1; w < WSX_MAX_STREAMS; w++) {
        if (c->aux[w]) (void)hipStreamDestroy(c->aux[w]);
        if (c->ev_joins[w]) (void)hipEventDestroy(c->ev_joins[w]);
    }
    for (void *p : c->pinned)
        if (p) (void)hipHostFree(p);
    for (hipEvent_t e : c->ev_meta)
        if (e) (void)hipEventDestroy(e);
    if (c->ev_joins[0]) (void)hipEventDestroy(c->ev_joins[0]);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);

    if (c->ev_window) (void)hipEventDestroy(c->ev_window);
    if (c->ev_begin) (void)hipEventDestroy(c->ev_begin);
    if (c->ev_end) (void)hipEventDestroy(c->ev_end);
    for (auto &e : c->sched_events) (void)hipEventDestroy(e);
    for (auto &p : c->dp_events) {
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    delete c;
}

int wsx_caller_set_workspace_limit(wsx_caller *c, uint64_t bytes)
{
    if (!c || bytes < (64ull << 20)) return WSX_ERR_INVALID;
    c->ws_limit = bytes;
    return WSX_SUCCESS;
}

int wsx_caller_get_workspace_limit(wsx_caller *c, uint64_t *bytes)
{
    if (!c || !bytes) return WSX_ERR_INVALID;
    *bytes = c->ws_limit;
    return WSX_SUCCESS;
}

int wsx_caller_set_tuning(wsx_caller *c, int32_t knob, int64_t value)
{
    if (!c) return WSX_ERR_INVALID;
    switch (knob) {
    case WSX_TUNE_STREAM_TRACEBACK_MIN: c->tun.stream_traceback_min = (int32_t)std::max<int64_t>(1, std::min<int64_t>(value, INT32_MAX)); break;
    case WSX_TUNE_BORDERS_WAVE_BELOW: c->tun.borders_wave_below = (int32_t)std::max<int64_t>(0, std::min<int64_t>(value, INT32_MAX)); break;
    case WSX_TUNE_SEGMENT_TWO_KERNELS: c->tun.segment_two_kernels = value != 0; break;
    case WSX_TUNE_FILL_BLOCKS_PER_CU: c->tun.fill_blocks_per_cu = (int32_t)std::max<int64_t>(0, std::min<int64_t>(value, 64)); break;
    case WSX_TUNE_CHUNKS: c->chunks_override = (int)std::max<int64_t>(0, std::min<int64_t>(value, 4096)); break;
    case WSX_TUNE_SMALL_PIPE_SAMPLES: c->small_pipe_samples = std::max<int64_t>(0, value); break;
    case WSX_TUNE_CALLS_IN_FLIGHT: c->in_flight = (int)std::max<int64_t>(2, std::min<int64_t>(value, wsx_caller::kMetaSlots)); break;
    case WSX_TUNE_SMALL_CALLS_IN_FLIGHT: c->in_flight_small = (int)std::max<int64_t>(2, std::min<int64_t>(value, wsx_caller::kMetaSlots)); break;
    default: g_err = "wsx_caller_set_tuning: unknown knob"; return WSX_ERR_INVALID;
    }
    return WSX_SUCCESS;
}


int wsx_caller_set_streams(wsx_caller *c, int32_t n_streams)
{
    if (!c || n_streams < 1 || n_streams > WSX_MAX_STREAMS) return WSX_ERR_INVALID;
    HIPCHK(hipSetDevice(c->device));
    c->n_streams = n_streams; // (streams are created on first use)
    return WSX_SUCCESS;
}

int wsx_caller_synchronize(wsx_caller *c)
{
    if (!c) return WSX_ERR_INVALID;
    HIPCHK(hipSetDevice(c->device));
    if (c->join_st) HIPCHK(hipStreamSynchronize(c->join_st)); // pipelined calls end there
    HIPCHK(hipStreamSynchronize(c->stream));
    return WSX_SUCCESS;
}

int wsx_caller_join(wsx_caller *c, void *stream)
{
    if (!c) return WSX_ERR_INVALID;
    HIPCHK(hipSetDevice(c->device));
    // a never-recorded event is complete: joining before the first call is a no-op
    HIPCHK(hipStreamWaitEvent(stream ? (hipStream_t)stream : c->stream, c->ev_end, 0));
    return WSX_SUCCESS;
}

int wsx_caller_timing_window(wsx_caller *c, int32_t on)
{
    if (!c) return WSX_ERR_INVALID;
    HIPCHK(hipSetDevice(c->device));
    if (on && !c->ev_window) HIPCHK(hipEventCreate(&c->ev_window));
    c->timing_window = on != 0;
    c->window_open = false;
    c->dp_events_used = 0;
    c->timing_valid = false;
    return WSX_SUCCESS;
}

int wsx_caller_set_pipelined(wsx_caller *c, int32_t on)
{
    if (!c) return WSX_ERR_INVALID;
    HIPCHK(hipSetDevice(c->device));
    if (on && !c->join_st) HIPCHK(hipStreamCreateWithFlags(&c->join_st, hipStreamNonBlocking));
    if (!on && c->pipelined) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_end, 0)); // back to stream order
    c->pipelined = on != 0;
    return WSX_SUCCESS;
}

const char *wsx_caller_kernel_name(wsx_caller *c, int32_t a)
{
    if (!c || a < 0 || a >= (int)c->variant.size()) return "";
    const Variant &v = c->variant[a];
    return wsx_pass_kernel_name(c->prm.min_values_per_state, v.K, v.F, v.FL, v.pk, v.lm, v.generic);
}

} // extern "C"

// accessors for the other translation units of the library (wsx_prep.hip)
int wsx_internal_device(wsx_caller *c) { return c->device; }
hipStream_t wsx_internal_stream(wsx_caller *c) { return c->stream; }
void wsx_internal_set_error(const char *msg) { g_err = msg; }
// No C++ exception crosses the C ABI: the entry points that allocate host memory are function-try-blocks ending here.
int wsx_internal_on_exception(void)
{
    try {
        throw;
    } catch (const std::bad_alloc &) {
        g_err = "out of host memory";
        return WSX_ERR_NOMEM;
    } catch (const std::exception &e) {
        g_err = std::string("internal error: ") + e.what();
        return WSX_ERR_INVALID;
    } catch (...) {
        g_err = "internal error (unknown exception)";
        return WSX_ERR_INVALID;
    }
}
uint64_t wsx_internal_workspace_limit(wsx_caller *c) { return c->ws_limit; }
hipError_t wsx_internal_prep_pinned(wsx_caller *c, size_t bytes, void **p, hipEvent_t *last_use)
{
    const int k = (int)(c->prep_turn++ % wsx_caller::kPrepSlots);
    hipError_t e = hipSuccess;
    if (!c->ev_prep[k] && (e = hipEventCreateWithFlags(&c->ev_prep[k], hipEventDisableTiming)) != hipSuccess) return e;
    if (bytes > c->prep_pinned_cap[k]) {
        if ((e = hipEventSynchronize(c->ev_prep[k])) != hipSuccess) return e; // uploads from the old buffer
        if (c->prep_pinned[k]) (void)hipHostFree(c->prep_pinned[k]);
        c->prep_pinned[k] = nullptr;
        c->prep_pinned_cap[k] = 0;
        if ((e = hipHostMalloc(&c->prep_pinned[k], bytes + bytes / 4, hipHostMallocDefault)) != hipSuccess) return e;
        c->prep_pinned_cap[k] = bytes + bytes / 4;
    }
    *p = c->prep_pinned[k];
    *last_use = c->ev_prep[k];
    return hipSuccess;
}
// the next slot of the ring wsx_vbz_decode stages its block tables in (csrc/wsx_vbz.hip)
hipError_t wsx_internal_vbz_slot(wsx_caller *c, size_t bytes, void **host, void **dev, hipEvent_t *last_use)
{
    auto &v = c->vbz_ring[c->vbz_turn++ % 8];
    hipError_t e = hipSuccess;
    if (!v.ev && (e = hipEventCreateWithFlags(&v.ev, hipEventDisableTiming)) != hipSuccess) return e;
    if (bytes > v.host_cap) {
        if ((e = hipEventSynchronize(v.ev)) != hipSuccess) return e;
        if (v.host) (void)hipHostFree(v.host);
        v.host = nullptr;
        v.host_cap = 0;
        if ((e = hipHostMalloc(&v.host, bytes + bytes / 2, hipHostMallocDefault)) != hipSuccess) return e;
        v.host_cap = bytes + bytes / 2;
    }
    if ((e = v.dev.ensure(bytes + bytes / 2)) != hipSuccess) return e;
    *host = v.host;
    *dev = v.dev.p;
    *last_use = v.ev;
    return hipSuccess;
}
// wsx_zstd_decode's own status array (a caller that passes none still needs one between its two kernels): grown when needed -- after
// the stream has drained: an earlier call's kernels may still use it --, freed with the handle
hipError_t wsx_internal_zstd_status(wsx_caller *c, size_t bytes, void **p)
{
    if (bytes > c->zstd_status.cap) {
        hipError_t e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) return e;
    }
    hipError_t e = c->zstd_status.ensure(bytes);
    *p = c->zstd_status.p;
    return e;
}
// buffer `slot` of the signal loader's pool, at least `bytes` large (grown when needed, freed with the handle)
hipError_t wsx_internal_prep_buffer(wsx_caller *c, int slot, size_t bytes, void **p)
{
    hipError_t e = c->prep_pool[slot].ensure(bytes);
    *p = c->prep_pool[slot].p;
    return e;
}

namespace {

// Common driver for wsx_call_batch (full = true) and wsx_warp_batch (full = false).
struct BatchIO {
    int mem;
    const double *signal;
    const int64_t *offsets;
    const int32_t *aut_id;
    int64_t n;
    const double *const *read_ptrs = nullptr; // host reads in separate arrays (instead of `signal`)
    // full call
    wsx_result *results = nullptr;
    wsx_traces traces{};
    // warp only
    const uint8_t *mask = nullptr;
    uint16_t *trace = nullptr;
    double *end_cost = nullptr;
    double *last_row = nullptr;
    int32_t last_row_stride = 0;
    int32_t *status = nullptr;
};

// ---- a call in three parts: validate_batch (the caller's arguments), plan_chunks (which reads go together: host logic
// only, no HIP call), run_batch (workspace, per-chunk pointers, the launch sequence on the handle's streams) ----------------
int validate_batch(wsx_caller *c, const BatchIO &io, bool full)
{
    if (!c || !io.offsets || !io.aut_id || io.n < 0 || (io.n > 0 && !io.signal && !io.read_ptrs)) {
        g_err = "null argument";
        return WSX_ERR_INVALID;
    }
    if (io.mem != WSX_MEM_HOST && io.mem != WSX_MEM_DEVICE) {
        g_err = "mem must be WSX_MEM_HOST or WSX_MEM_DEVICE";
        return WSX_ERR_INVALID;
    }
    if (full && !io.results) {
        g_err = "results is NULL";
        return WSX_ERR_INVALID;
    }
    if (!full && !io.trace) {
        g_err = "trace is NULL";
        return WSX_ERR_INVALID;
    }
    const int nA = (int)c->variant.size();
    for (int64_t r = 0; r < io.n; r++) {
        const int64_t T = io.offsets[r + 1] - io.offsets[r];
        if (T < 0 || T > (1 << 30)) {
            g_err = "offsets must be non-decreasing (read too long or negative length)";
            return WSX_ERR_INVALID;
        }
        if (io.aut_id[r] < 0 || io.aut_id[r] >= nA) {
            g_err = "automaton_id out of range";
            return WSX_ERR_INVALID;
        }
    }
    if (full && (io.traces.seq1 || io.traces.seq2) && !c->have_bases) {
        g_err = "sequences requested but an automaton was created without last_base";
        return WSX_ERR_INVALID;
    }
    return WSX_SUCCESS;
}

// Which reads go together.  `pipe`: a pipelined device-buffer call; small_pipe_samples: up to this many samples such a
// call stays in one chunk (four calls side by side).  psb / prb: workspace bytes per sample / per read.
std::vector<ChunkPlan> plan_chunks(const wsx_caller *c, const BatchIO &io, bool pipe, int64_t small_pipe_samples, size_t psb, size_t prb)
{
    const int64_t n = io.n;
    // The limit covers everything the call allocates: up to streams_per_call work sets exist side by side, and a buffer
    // grows with 12.5 % headroom (DeviceBuf) -- so one chunk may take limit / sets / 1.125.
    const size_t sets = (size_t)std::max(1, std::min(c->n_streams, c->streams_per_call));
    const size_t chunk_limit = (size_t)((double)c->ws_limit / (double)sets / 1.125);
    std::vector<ChunkPlan> chunks;
    auto chunk_need = [&](int64_t samples, int64_t reads) { return (size_t)samples * psb + (size_t)reads * prb; };
    auto greedy_plan = [&]() { // as many reads per chunk as the limit allows (ragged batches, or when the even split fails)
        chunks.clear();
        int64_t first = 0;
        while (first < n) {
            int64_t cnt = 0, smp = 0;
            int mt = 0;
            while (first + cnt < n) {
                const int64_t T = io.offsets[first + cnt + 1] - io.offsets[first + cnt];
                if (cnt > 0 && chunk_need(smp + T, cnt + 1) > chunk_limit) break;
                smp += T;
                cnt++;
                mt = std::max<int>(mt, (int)T);
            }
            chunks.push_back({first, cnt, io.offsets[first], smp, mt});
            first += cnt;
        }
    };
    // How many chunks: big batches are split so that chunks can overlap on the streams (the latency-bound stages of one
    // chunk run under the VALU-bound fill of another).  The thread-per-read stages want launches of ~25k reads, so a
    // second round of chunks per stream only pays from ~200k reads on, or when reads are long (their serial stages then
    // last long enough to need another chunk's fill to hide under); measured in profiles/r01s5_chunk_sweep.log.  Small
    // pipelined calls: one chunk, consecutive calls on consecutive streams -- four calls side by side fill the chip better
    // than one call cut into four (profiles/r02_small_call_sweep.log).  WSX_CHUNKS (read when the handle is
    // created) overrides the count.  The workspace limit may ask for more chunks than that.
    const int spc = std::min(c->n_streams, c->streams_per_call);
    const int64_t total_samples = io.offsets[n] - io.offsets[0];
    const bool small_call = total_samples < (int64_t)80 << 20;
    int want = 1;
    if (n >= 4096) {
        want = (pipe && small_call) ? ((c->in_flight_small > 2 && total_samples <= small_pipe_samples) ? 1 : 2) : (n >= 8192 ? spc : 2);
        if (n >= 32768) {
            const bool long_reads = total_samples / n >= 4096;
            const int64_t per_round = (int64_t)25000 * spc;
            want = spc * ((long_reads || 2 * n >= 3 * per_round) ? 2 : 1);
        }
        if (spc == 1) want = 1;
        if (c->chunks_override > 0) want = c->chunks_override;
    }
    {
        const size_t by_limit = (chunk_need(total_samples, n) + chunk_limit - 1) / chunk_limit;
        if (by_limit > (size_t)want) want = (int)((by_limit + spc - 1) / spc * spc); // whole rounds of the streams
        want = (int)std::min<int64_t>(want, n);
        int64_t first = 0;
        bool fits = true;
        for (int part = 0; part < want && first < n; part++) {
            const int64_t target = total_samples * (part + 1) / want; // cumulative samples at the end of this part
            int64_t cnt = 0;
            int mt = 0;
            while (first + cnt < n && (io.offsets[first + cnt] - io.offsets[0] < target || part == want - 1)) {
                mt = std::max<int>(mt, (int)(io.offsets[first + cnt + 1] - io.offsets[first + cnt]));
                cnt++;
            }
            if (cnt == 0) continue;
            const int64_t smp = io.offsets[first + cnt] - io.offsets[first];
            fits = fits && (cnt == 1 || chunk_need(smp, cnt) <= chunk_limit);
            chunks.push_back({first, cnt, io.offsets[first], smp, mt});
            first += cnt;
        }
        if (!fits) greedy_plan();
    }
    return chunks;
}

int run_batch(wsx_caller *c, const BatchIO &io, bool full)
{
    int rc = validate_batch(c, io, full);
    if (rc) return rc;
    const int64_t n = io.n;
    const int nA = (int)c->variant.size();
    rc = set_device(c);
    if (rc) return rc;
    c->timing_valid = false;
    if (!c->timing_window) c->dp_events_used = 0;
    if (n == 0) return WSX_SUCCESS;
    hipStream_t st = c->stream;
    const bool host = io.mem == WSX_MEM_HOST;
    // Pipelined (device buffers only): the call does not join the handle's stream at its end, so the next call's
    // chunks follow this call's on every internal stream without a gap; wsx_caller_join orders a consumer after it.
    const bool pipe = c->pipelined && !host;
    if (c->pipelined && host) HIPCHK(hipStreamWaitEvent(st, c->ev_end, 0));
    // The host may run this many pipelined calls ahead of the device: two for batches that fill the chip by themselves, up
    // to kMetaSlots for small ones (each takes one stream, consecutive calls rotate over the streams: four side by side;
    // 12 500 reads: 1.96-2.02 ms per call against 2.10-2.14 with two calls of two chunks each, profiles/r02_small_call_sweep.log)
    // (up to 50 M samples per call; at 60 M -- 30 000 reads of 2 000 samples, 20 000 of 3 000 -- four whole calls side by
    // side ran at half the speed of two calls of two chunks each: the bound below keeps to what was measured)
    const int64_t small_pipe_samples = c->small_pipe_samples;
    const bool small_pipe = pipe && n > 0 && (io.offsets[n] - io.offsets[0]) <= small_pipe_samples && n < 32768;
    const int depth = small_pipe ? c->in_flight_small : c->in_flight;
    const uint64_t seq = pipe ? c->call_seq++ : 0;
    const int slot = pipe ? (int)(seq % (uint64_t)wsx_caller::kMetaSlots) : 0;
    const bool want_traces = full && (io.traces.trace1 || io.traces.trace2 || io.traces.rescaled || io.traces.badmask ||
                                      io.traces.seq1 || io.traces.seq2);

    // ---- metadata on the device: offsets, automaton ids, launch order ---------------------------
    // order: reads grouped by DP kernel variant, longest first inside a group (load balance)
    // the call that used this slot last (the previous one; in pipelined mode the one before that) has to be over:
    // its kernels read the device copy, its uploads the pinned one (no-op if never recorded)
    HIPCHK(hipEventSynchronize(c->ev_meta[slot]));
    if (pipe && depth < wsx_caller::kMetaSlots && seq >= (uint64_t)depth)
        HIPCHK(hipEventSynchronize(c->ev_meta[(seq - depth) % (uint64_t)wsx_caller::kMetaSlots]));
    HIPCHK(c->meta[slot].ensure(align_up((n + 1) * 8) + 2 * align_up(n * 4) + align_up(n * 8)));
    if (pipe) // (the other slots too, once: an allocation in the middle of a pipelined sequence stalls the streams)
        for (int q = 0; q < wsx_caller::kMetaSlots; q++) {
            const size_t need = align_up((n + 1) * 8) + 2 * align_up(n * 4) + align_up(n * 8);
            if (q == slot || (c->meta[q].cap >= need && c->pinned_cap[q] >= need)) continue;
            HIPCHK(hipEventSynchronize(c->ev_meta[q])); // (only ever waits when the batch size grows mid-sequence)
            HIPCHK(c->meta[q].ensure(need));
            if (need > c->pinned_cap[q]) {
                if (c->pinned[q]) (void)hipHostFree(c->pinned[q]);
                c->pinned[q] = nullptr;
                c->pinned_cap[q] = 0;
                HIPCHK(hipHostMalloc(&c->pinned[q], need + need / 4, hipHostMallocDefault));
                c->pinned_cap[q] = need + need / 4;
            }
        }
    Carver mc(c->meta[slot].p);
    int64_t *d_offsets = mc.take<int64_t>(n + 1);
    int32_t *d_autid = mc.take<int32_t>(n);
    int32_t *d_order = mc.take<int32_t>(n);
    int64_t *d_bpoff = mc.take<int64_t>(n); // per read (global index): start of its back-pointer rows in its chunk's region
    // metadata goes through a pinned buffer owned by the handle: the uploads are then truly asynchronous and the
    // caller's arrays are not referenced after this function returns
    const size_t pin_bytes = align_up((n + 1) * 8) + 2 * align_up(n * 4) + align_up(n * 8);
    if (pin_bytes > c->pinned_cap[slot]) {
        if (c->pinned[slot]) (void)hipHostFree(c->pinned[slot]);
        c->pinned[slot] = nullptr;
        c->pinned_cap[slot] = 0;
        HIPCHK(hipHostMalloc(&c->pinned[slot], pin_bytes + pin_bytes / 4, hipHostMallocDefault));
        c->pinned_cap[slot] = pin_bytes + pin_bytes / 4;
    }
    Carver pc(c->pinned[slot]);
    int64_t *h_offsets = pc.take<int64_t>(n + 1);
    int32_t *h_autid = pc.take<int32_t>(n);
    int32_t *h_order = pc.take<int32_t>(n);
    int64_t *h_bpoff = pc.take<int64_t>(n);
    memcpy(h_offsets, io.offsets, (n + 1) * 8);
    memcpy(h_autid, io.aut_id, n * 4);
    HIPCHK(hipMemcpyAsync(d_offsets, h_offsets, (n + 1) * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_autid, h_autid, n * 4, hipMemcpyHostToDevice, st));

    // ---- chunk plan -----------------------------------------------------------------------------
    const size_t psb = per_sample_bytes(c, host, want_traces || !full);
    const size_t prb = per_read_bytes(c, host, io.last_row ? (size_t)std::max(io.last_row_stride, 0) * 8 : 0);
    const std::vector<ChunkPlan> chunks = plan_chunks(c, io, pipe, small_pipe_samples, psb, prb);
    const int spc = std::min(c->n_streams, c->streams_per_call);
    const bool small_call = (io.offsets[n] - io.offsets[0]) < (int64_t)80 << 20;
    size_t max_smp = 0, max_cnt = 0;
    for (auto &ch : chunks) {
        max_smp = std::max<size_t>(max_smp, ch.samples);
        max_cnt = std::max<size_t>(max_cnt, ch.count);
    }
    const size_t S1 = max_smp + 64; // per-sample capacity
    const size_t R1 = max_cnt + 8;
    // per-sample arrays
    size_t smp_bytes = align_up(S1 * 8) /*rescaled*/ + align_up(S1 * 2) + align_up(S1 * 4) /*runs*/ +
                       3 * align_up(S1 * 8) + align_up(S1) /*alignment*/ + 4 * align_up(S1 * 8) /*fit + scratch*/ +
                       align_up((S1 / 32 + R1 + 2) * 4) /*mask bits*/;
    const int n_work = (int)std::min<size_t>(chunks.size(), (size_t)spc);
    // Work set (and stream) of chunk ci: (rot + ci) mod n_streams.  A pipelined call with fewer chunks than the handle has
    // streams starts where the previous call ended, so that two back-to-back calls run on disjoint streams side by side
    // (small batches: one call's kernels are too few wavefronts to fill the chip, and its stages depend on each other).
    // (big calls fill the chip by themselves and stay on the first streams_per_call sets: measured in
    // profiles/r02_stream_share_sweep.log)
    const int rot = (pipe && small_call && n_work < c->n_streams) ? c->rot % c->n_streams : 0;
    if (pipe) c->rot = (rot + n_work) % c->n_streams;
    auto wset = [&](size_t ci) -> int { return (rot + (int)(ci % n_work)) % c->n_streams; };
    // A pipelined call that rotates over the streams sizes the work sets of ALL of them, not only its own: the next calls
    // land on the other sets, and a multi-gigabyte hipMalloc in the middle of a pipelined sequence stalls every stream
    // (seen as 8 instead of 4.6 ms per step over 30 steps of 30 000 reads, 26-33 instead of 11.5 ms for the flank-110 shape,
    // when the benchmark's two warm-up calls had touched only two of the four sets: profiles/r02_small_call_sweep.log).
    const bool rotates = pipe && small_call && n_work < c->n_streams;
    const int n_sized = rotates ? c->n_streams : n_work;
    auto sized_set = [&](int k) -> int { return rotates ? k : wset(k); };
    for (int k = 0; k < n_sized; k++) {
        const int w = sized_set(k);
        if (w > 0 && !c->aux[w]) {
            HIPCHK(hipStreamCreateWithFlags(&c->aux[w], hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&c->ev_joins[w], hipEventDisableTiming));
        }
        HIPCHK(c->work[w].samples.ensure(smp_bytes));
        HIPCHK(c->work[w].reads.ensure(R1 * 168 + align_up(R1 * sizeof(wsx_result)) + 8192));
        if (full && c->prm.reps_as_one)
            HIPCHK(c->work[w].reps.ensure(R1 * 2 * (size_t)c->max_states * sizeof(int32_t)));
    }
    // back-pointer scratch: the reads of a chunk lie back to back in one region, each with the rows its own kernel variant
    // writes (h_bpoff: 64-bit words from the region's start) -- launch groups of a chunk never share words, their fills and
    // tracebacks may be issued in any order, and a handle with many variants (mixed loci) needs no more than its reads do.
    // The tracebacks fetch whole blocks of rows and may look past a read's last row: the region ends with slack for that.
    // Launch order of every chunk first (host only): reads grouped by kernel variant, longest first inside a group (load
    // balance).
    struct Launches {
        std::vector<std::vector<int32_t>> groups;
        std::vector<Variant> gvar;
        std::vector<size_t> gpos;
    };
    std::vector<Launches> launches(chunks.size());
    constexpr size_t kBpSlackWords = 4096;
    size_t bp_words64 = 0;
    for (size_t ci = 0; ci < chunks.size(); ci++) {
        const ChunkPlan &ch = chunks[ci];
        Launches &L = launches[ci];
        for (int64_t r = ch.first; r < ch.first + ch.count; r++) {
            const Variant &v = c->variant[io.aut_id[r]];
            size_t g = 0;
            for (; g < L.gvar.size(); g++)
                if (L.gvar[g].same(v)) break;
            if (g == L.gvar.size()) {
                L.gvar.push_back(v);
                L.groups.emplace_back();
            }
            L.groups[g].push_back((int32_t)r);
        }
        size_t pos = (size_t)ch.first, at = 0;
        for (size_t g = 0; g < L.groups.size(); g++) {
            auto &grp = L.groups[g];
            std::stable_sort(grp.begin(), grp.end(), [&](int32_t p, int32_t q) {
                return (io.offsets[p + 1] - io.offsets[p]) > (io.offsets[q + 1] - io.offsets[q]);
            });
            L.gpos.push_back(pos);
            std::copy(grp.begin(), grp.end(), h_order + pos);
            pos += grp.size();
            const Variant &v = L.gvar[g];
            for (int32_t r : grp) {
                h_bpoff[r] = (int64_t)at;
                at += v.bp_read_words((size_t)(io.offsets[r + 1] - io.offsets[r]));
            }
        }
        bp_words64 = std::max(bp_words64, at + kBpSlackWords);
    }
    HIPCHK(hipMemcpyAsync(d_bpoff, h_bpoff, (size_t)n * 8, hipMemcpyHostToDevice, st));
    for (int k = 0; k < n_sized; k++) HIPCHK(c->work[sized_set(k)].bp.ensure(bp_words64 * 8));
    for (int k = 0; k < n_work && host; k++) {
        const int w = wset(k);
        HIPCHK(c->work[w].stage_sig.ensure(S1 * 8));
        size_t so = align_up(S1 * 2) * 2 + align_up(S1 * 8) + 3 * align_up(S1) + align_up(R1 * 8) + align_up(R1 * 4) +
                    align_up(R1 * (size_t)std::max(io.last_row_stride, 1) * 8);
        HIPCHK(c->work[w].stage_out.ensure(so));
    }

    const hipStream_t main_st = c->stream;
    HIPCHK(hipEventRecord(c->ev_begin, main_st));
    if (c->timing_window && !c->window_open) {
        HIPCHK(hipEventRecord(c->ev_window, main_st));
        c->window_open = true;
    }
    const int m = c->prm.min_values_per_state;
    int32_t *order = h_order;

    // ---- per-chunk context: every device pointer, the launch groups, the stage arguments -------------------------
    struct Ctx {
        ChunkPlan ch;
        wsx_caller::Work *W;
        const double *d_sig;
        double *d_resc, *d_resc_user, *d_endcost, *d_endcost_user, *d_lastrow, *d_coef;
        int32_t *d_smooth_cnt;
        uint16_t *d_tr1, *d_tr2;
        uint8_t *d_badmask, *d_seq1, *d_seq2, *d_alg;
        int32_t *d_status, *d_status_user, *d_nruns;
        uint32_t *d_maskbits;
        wsx_result *d_results;
        std::vector<std::vector<int32_t>> groups;
        std::vector<Variant> gvar;
        std::vector<size_t> gpos;
        PassArgs pa;
        MidArgs ma;
        FitArgs fa;
        EvalArgs ea;
    };
    std::vector<Ctx> ctxs(chunks.size());

    // Carve the work set of chunk ci, upload its launch order (and, for host buffers, its signal) on stream s.
    // caller-owned host buffers: big batches go through the pinned rings, small ones use plain (blocking) copies
    const bool ringed = host && ((size_t)(io.offsets[n] - io.offsets[0]) * 8 >= (16u << 20) || io.read_ptrs);
    wsx_result *res_dst = io.results;
    if (ringed) {
        HIPCHK(c->ring_up.init(4));
        HIPCHK(c->ring_down.init(8));
        c->ring_up.discard();
        c->ring_down.discard();
        if (full) {
            const size_t need = (size_t)n * sizeof(wsx_result);
            if (need > c->pinned_res_cap) {
                if (c->pinned_res) (void)hipHostFree(c->pinned_res);
                c->pinned_res = nullptr;
                c->pinned_res_cap = 0;
                HIPCHK(hipHostMalloc(&c->pinned_res, need + need / 4, hipHostMallocDefault));
                c->pinned_res_cap = need + need / 4;
            }
            res_dst = (wsx_result *)c->pinned_res;
        }
    }
    auto h2d = [&](void *dst, const void *src, size_t bytes, hipStream_t s) -> hipError_t {
        return ringed ? c->ring_up.upload(dst, src, bytes, s) : hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s);
    };
    auto d2h = [&](void *dst, const void *src, size_t bytes, hipStream_t s) -> hipError_t {
        return ringed ? c->ring_down.download(dst, src, bytes, s) : hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s);
    };

    auto prepare = [&](size_t ci, hipStream_t s) -> int {
        Ctx &x = ctxs[ci];
        x.ch = chunks[ci];
        x.W = &c->work[wset(ci)];
        wsx_caller::Work &W = *x.W;
        const int64_t f = x.ch.first, cnt = x.ch.count, boff = x.ch.base_off;
        Carver sc(W.samples.p);
        x.d_resc = sc.take<double>(S1);
        uint16_t *d_run_state = sc.take<uint16_t>(S1);
        int32_t *d_run_start = sc.take<int32_t>(S1);
        double *d_alv = sc.take<double>(S1), *d_ale = sc.take<double>(S1), *d_alc = sc.take<double>(S1);
        x.d_alg = sc.take<uint8_t>(S1);
        double *d_fx = sc.take<double>(S1), *d_fy = sc.take<double>(S1);
        double *d_scr0 = sc.take<double>(S1), *d_scr1 = sc.take<double>(S1);
        // the t-statistics of the two-kernel segmentation (reads beyond ~90 k samples) live where the rescaled signal will
        // be: that stage runs in the first pass only, before eval_kernel writes the plane, and nothing reads it in between
        double *d_scr2 = x.d_resc;
        x.d_maskbits = sc.take<uint32_t>(S1 / 32 + R1 + 2);
        Carver rcv(W.reads.p);
        x.d_nruns = rcv.take<int32_t>(R1);
        x.d_status = rcv.take<int32_t>(R1);
        int32_t *d_fitm = rcv.take<int32_t>(R1);
        x.d_endcost = rcv.take<double>(R1);
        x.d_coef = rcv.take<double>(R1 * 6);
        MidRec *d_rec = rcv.take<MidRec>(R1);
        int32_t *d_nalign = rcv.take<int32_t>(R1);
        wsx_result *d_results_ws = rcv.take<wsx_result>(R1);
        int32_t *d_smooth_list = rcv.take<int32_t>(R1), *d_smooth_slot = rcv.take<int32_t>(R1);
        x.d_smooth_cnt = rcv.take<int32_t>(4);

        if (host) {
            if (io.read_ptrs) HIPCHK(c->ring_up.upload_gather(W.stage_sig.p, io.read_ptrs, io.offsets, f, cnt, s));
            else HIPCHK(h2d(W.stage_sig.p, io.signal + boff, (size_t)x.ch.samples * 8, s));
            x.d_sig = (const double *)W.stage_sig.p;
        } else {
            x.d_sig = io.signal + boff;
        }
        // user-visible per-sample / per-read outputs of this chunk (device pointers)
        Carver oc(host ? W.stage_out.p : nullptr);
        x.d_tr1 = x.d_tr2 = nullptr;
        x.d_resc_user = x.d_endcost_user = x.d_lastrow = nullptr;
        x.d_badmask = x.d_seq1 = x.d_seq2 = nullptr;
        x.d_status_user = nullptr;
        x.d_results = nullptr;
        if (full) {
            if (host) {
                if (io.traces.trace1) x.d_tr1 = oc.take<uint16_t>(S1);
                if (io.traces.trace2) x.d_tr2 = oc.take<uint16_t>(S1);
                if (io.traces.rescaled) x.d_resc_user = oc.take<double>(S1);
                if (io.traces.badmask) x.d_badmask = oc.take<uint8_t>(S1);
                if (io.traces.seq1) x.d_seq1 = oc.take<uint8_t>(S1);
                if (io.traces.seq2) x.d_seq2 = oc.take<uint8_t>(S1);
                x.d_results = d_results_ws;
            } else {
                x.d_tr1 = io.traces.trace1 ? io.traces.trace1 + boff : nullptr;
                x.d_tr2 = io.traces.trace2 ? io.traces.trace2 + boff : nullptr;
                x.d_resc_user = io.traces.rescaled ? io.traces.rescaled + boff : nullptr;
                x.d_badmask = io.traces.badmask ? io.traces.badmask + boff : nullptr;
                x.d_seq1 = io.traces.seq1 ? io.traces.seq1 + boff : nullptr;
                x.d_seq2 = io.traces.seq2 ? io.traces.seq2 + boff : nullptr;
                x.d_results = io.results + f;
            }
        } else {
            if (host) {
                x.d_tr1 = oc.take<uint16_t>(S1);
                if (io.end_cost) x.d_endcost_user = oc.take<double>(R1);
                if (io.status) x.d_status_user = oc.take<int32_t>(R1);
                if (io.last_row) x.d_lastrow = oc.take<double>(R1 * (size_t)io.last_row_stride);
            } else {
                x.d_tr1 = io.trace + boff;
                x.d_endcost_user = io.end_cost ? io.end_cost + f : nullptr;
                x.d_status_user = io.status ? io.status + f : nullptr;
                x.d_lastrow = io.last_row ? io.last_row + (size_t)f * io.last_row_stride : nullptr;
            }
        }

        // launch order of this chunk (built above): group by kernel variant, longest first
        x.groups = std::move(launches[ci].groups);
        x.gvar = std::move(launches[ci].gvar);
        x.gpos = std::move(launches[ci].gpos);
        HIPCHK(hipMemcpyAsync(d_order + f, order + f, (size_t)cnt * 4, hipMemcpyHostToDevice, s));

        PassArgs &pa = x.pa;
        pa = PassArgs{};
        pa.aut = (const DevAutomaton *)c->aut_table.p;
        pa.offsets = d_offsets;
        pa.aut_id = d_autid;
        pa.first_read = (int32_t)f;
        pa.base_off = boff;
        pa.bp = (uint32_t *)W.bp.p;
        pa.bp_off = d_bpoff + f;
        pa.run_state = d_run_state;
        pa.run_start = d_run_start;
        pa.n_runs = x.d_nruns;
        pa.last_row_stride = io.last_row_stride;
        pa.m = m;

        MidArgs &ma = x.ma;
        ma = MidArgs{};
        ma.aut = (const DevAutomaton *)c->aut_table.p;
        ma.prm = DevParams{c->prm.min_values_per_state, c->prm.states_in_segment, c->prm.threshold, c->prm.max_std,
                           c->prm.method_median, c->prm.reps_as_one};
        ma.offsets = d_offsets;
        ma.aut_id = d_autid;
        ma.n_reads = (int32_t)cnt;
        ma.first_read = (int32_t)f;
        ma.base_off = boff;
        ma.run_state = d_run_state;
        ma.run_start = d_run_start;
        ma.n_runs = x.d_nruns;
        ma.al_value = d_alv;
        ma.al_expected = d_ale;
        ma.al_cost = d_alc;
        ma.al_good = x.d_alg;
        ma.fit_x = d_fx;
        ma.fit_y = d_fy;
        ma.fit_m = d_fitm;
        ma.rec = d_rec;
        ma.n_align = c->prm.reps_as_one ? d_nalign : nullptr;
        ma.state_scratch = (int32_t *)W.reps.p;
        ma.max_states = c->max_states;
        ma.scr0 = d_scr0;
        ma.scr1 = d_scr1;
        ma.scr2 = d_scr2;
        ma.status = x.d_status;
        ma.end_cost = x.d_endcost;
        ma.results = x.d_results;
        const bool may_smooth = c->prm.threshold > 1.0; // FITPACK can leave its polynomial branch only then
        x.fa = FitArgs{d_offsets, (int32_t)cnt, (int32_t)f, boff, d_fx, d_fy, d_fitm, x.d_coef, x.d_status,
                       may_smooth ? x.d_smooth_cnt : nullptr, may_smooth ? d_smooth_list : nullptr,
                       may_smooth ? d_smooth_slot : nullptr};
        x.ea = EvalArgs{d_offsets, (int32_t)cnt, (int32_t)f, boff, x.d_sig, x.d_coef, x.d_status, x.d_resc, x.d_resc_user,
                        nullptr, nullptr, 0};
        return WSX_SUCCESS;
    };

    // DP fill of one pass for every launch group of the chunk (timed with HIP events on the launching stream)
    auto do_fill = [&](Ctx &x, const double *sigp, const uint32_t *maskbits, int check_status, double *end_cost,
                       double *lastrow, int32_t *status, hipStream_t s) -> int {
        for (size_t g = 0; g < x.groups.size(); g++) {
            PassArgs pa = x.pa;
            pa.signal = sigp;
            pa.order = d_order + x.gpos[g];
            pa.n_launch = (int32_t)x.groups[g].size();
            pa.maskbits = maskbits;
            pa.end_cost = end_cost;
            pa.last_row = lastrow;
            pa.status = status;
            pa.check_status = check_status;
            hipEvent_t e0, e1;
            int rc2 = get_event_pair(c, &e0, &e1, pa.n_launch);
            if (rc2) return rc2;
            HIPCHK(hipEventRecord(e0, s));
            const Variant &lv = x.gvar[g];
            HIPCHK(wsx_launch_fill(pa, m, lv.K, lv.F, lv.FL, lv.pk, lv.lm, lv.generic, c->tun, s));
            HIPCHK(hipEventRecord(e1, s));
        }
        return WSX_SUCCESS;
    };
    auto do_traceback = [&](Ctx &x, const uint32_t *maskbits, uint16_t *trace, int32_t *status, hipStream_t s) -> int {
        for (size_t g = 0; g < x.groups.size(); g++) {
            PassArgs pa = x.pa;
            pa.order = d_order + x.gpos[g];
            pa.n_launch = (int32_t)x.groups[g].size();
            pa.maskbits = maskbits;
            pa.trace = trace;
            pa.status = status;
            const Variant &lv = x.gvar[g];
            pa.lane_major = lv.lm != 0;
            HIPCHK(wsx_launch_traceback(pa, lv.K, lv.F, lv.FL, lv.pk, lv.generic, nA, c->tun, s));
        }
        return WSX_SUCCESS;
    };
    // the four stages of a full call
    auto stage_f1 = [&](Ctx &x, hipStream_t s) -> int {
        return do_fill(x, x.d_sig, nullptr, 0, x.d_endcost, nullptr, x.d_status, s);
    };
    auto stage_m1 = [&](Ctx &x, hipStream_t s) -> int {
        int rc2 = do_traceback(x, nullptr, x.d_tr1, x.d_status, s);
        if (rc2) return rc2;
        MidArgs ma = x.ma;
        ma.signal = x.d_sig;
        ma.pass = 1;
        ma.maskbits = x.d_maskbits;
        ma.badmask_bytes = x.d_badmask;
        ma.seq_out = x.d_seq1;
        HIPCHK(wsx_launch_mid(ma, x.ch.max_T, c->tun, s));
        if (x.fa.smooth_list) HIPCHK(hipMemsetAsync(x.d_smooth_cnt, 0, 2 * sizeof(int32_t), s));
        HIPCHK(wsx_launch_fit(x.fa, s));
        if (x.fa.smooth_list) {
            // rescaling.threshold > 1: reads whose cubic fails fpcurf's test take FITPACK's knot-adding and smoothing
            // branch (a thread per read, arrays in a workspace sized from the chunk's count: the host has to see it)
            int32_t *hc = c->smooth_host + 2 * (x.W - c->work);
            HIPCHK(hipMemcpyAsync(hc, x.d_smooth_cnt, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            if (hc[0] > 0) {
                HIPCHK(x.W->smooth.ensure(wsx_smooth_workspace_bytes(hc[0], hc[1])));
                HIPCHK(wsx_launch_fit_smooth(x.fa, hc[0], hc[1], (double *)x.W->smooth.p, s));
                x.ea.smooth_slot = x.fa.smooth_slot;
                x.ea.smooth_ws = (const double *)x.W->smooth.p;
                x.ea.smooth_nest = hc[1] + 4 > 9 ? hc[1] + 4 : 9;
            }
        }
        HIPCHK(wsx_launch_eval(x.ea, x.ch.max_T, s));
        return WSX_SUCCESS;
    };
    auto stage_f2 = [&](Ctx &x, hipStream_t s) -> int {
        return do_fill(x, x.d_resc, x.d_maskbits, 1, x.d_endcost, nullptr, x.d_status, s);
    };
    auto stage_m2 = [&](Ctx &x, hipStream_t s) -> int {
        int rc2 = do_traceback(x, x.d_maskbits, x.d_tr2, x.d_status, s);
        if (rc2) return rc2;
        MidArgs ma = x.ma;
        ma.signal = x.d_resc;
        ma.pass = 2;
        ma.maskbits = nullptr;
        ma.badmask_bytes = nullptr;
        ma.seq_out = x.d_seq2;
        HIPCHK(wsx_launch_mid(ma, x.ch.max_T, c->tun, s));
        if (host) {
            const int64_t f = x.ch.first, cnt = x.ch.count, boff = x.ch.base_off;
            const size_t ns = (size_t)x.ch.samples;
            HIPCHK(hipMemcpyAsync(res_dst + f, x.d_results, (size_t)cnt * sizeof(wsx_result), hipMemcpyDeviceToHost, s));
            if (io.traces.trace1) HIPCHK(d2h(io.traces.trace1 + boff, x.d_tr1, ns * 2, s));
            if (io.traces.trace2) HIPCHK(d2h(io.traces.trace2 + boff, x.d_tr2, ns * 2, s));
            if (io.traces.rescaled) HIPCHK(d2h(io.traces.rescaled + boff, x.d_resc_user, ns * 8, s));
            if (io.traces.badmask) HIPCHK(d2h(io.traces.badmask + boff, x.d_badmask, ns, s));
            if (io.traces.seq1) HIPCHK(d2h(io.traces.seq1 + boff, x.d_seq1, ns, s));
            if (io.traces.seq2) HIPCHK(d2h(io.traces.seq2 + boff, x.d_seq2, ns, s));
        }
        return WSX_SUCCESS;
    };
    // the single stage of wsx_warp_batch (one DP + traceback, optional input mask)
    auto stage_warp = [&](Ctx &x, hipStream_t s) -> int {
        const int64_t f = x.ch.first, cnt = x.ch.count, boff = x.ch.base_off;
        const uint32_t *pass1_mask = nullptr;
        if (io.mask) {
            const uint8_t *d_mask_bytes;
            if (host) { // the alignment area doubles as staging for the mask bytes
                HIPCHK(hipMemcpyAsync(x.d_alg, io.mask + boff, (size_t)x.ch.samples, hipMemcpyHostToDevice, s));
                d_mask_bytes = x.d_alg;
            } else {
                d_mask_bytes = io.mask + boff;
            }
            dim3 grid((unsigned)cnt, (x.ch.max_T / 32 + 1 + 63) / 64);
            hipLaunchKernelGGL(pack_mask_kernel, grid, dim3(64), 0, s, d_mask_bytes, d_offsets, (int)f, boff, (int)cnt,
                               x.d_maskbits);
            HIPCHK(hipGetLastError());
            pass1_mask = x.d_maskbits;
        }
        int32_t *status = x.d_status_user ? x.d_status_user : x.d_status;
        int rc2 = do_fill(x, x.d_sig, pass1_mask, 0, x.d_endcost_user ? x.d_endcost_user : x.d_endcost, x.d_lastrow, status, s);
        if (rc2) return rc2;
        rc2 = do_traceback(x, pass1_mask, x.d_tr1, status, s);
        if (rc2) return rc2;
        if (host) {
            HIPCHK(d2h(io.trace + boff, x.d_tr1, (size_t)x.ch.samples * 2, s));
            if (io.end_cost) HIPCHK(hipMemcpyAsync(io.end_cost + f, x.d_endcost_user, (size_t)cnt * 8, hipMemcpyDeviceToHost, s));
            if (io.status) HIPCHK(hipMemcpyAsync(io.status + f, x.d_status_user, (size_t)cnt * 4, hipMemcpyDeviceToHost, s));
            if (io.last_row)
                HIPCHK(hipMemcpyAsync(io.last_row + (size_t)f * io.last_row_stride, x.d_lastrow,
                                      (size_t)cnt * io.last_row_stride * 8, hipMemcpyDeviceToHost, s));
        }
        return WSX_SUCCESS;
    };
    auto sched_event = [&](hipEvent_t *e) -> int {
        if (c->sched_used == c->sched_events.size()) {
            hipEvent_t ev;
            HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            c->sched_events.push_back(ev);
        }
        *e = c->sched_events[c->sched_used++];
        return WSX_SUCCESS;
    };
    c->sched_used = 0;

    {
        // ---- chunks rotate over the streams, each chunk's stages in order on its stream, staggered by one fill --------
        auto stream_of = [&](int w) -> hipStream_t { return w ? c->aux[w] : main_st; };
        if (n_work > 1 || rot != 0) {
            HIPCHK(hipEventRecord(c->ev_fork, main_st));
            for (int k = 0; k < n_work; k++)
                if (wset(k) != 0) HIPCHK(hipStreamWaitEvent(c->aux[wset(k)], c->ev_fork, 0));
        }
        for (size_t ci = 0; ci < chunks.size(); ci++) {
            Ctx &x = ctxs[ci];
            st = stream_of(wset(ci));
            // this work set's staging buffers were last used n_work chunks ago on the same stream (host copies)
            if (host && ci >= (size_t)n_work) HIPCHK(hipStreamSynchronize(st));
            if ((rc = prepare(ci, st))) return rc;
            if (!full) {
                if ((rc = stage_warp(x, st))) return rc;
                continue;
            }
            if ((rc = stage_f1(x, st))) return rc;
            if ((int)ci + 1 < n_work) { // stagger: the next stream starts after this chunk's first fill
                hipEvent_t e;
                if ((rc = sched_event(&e))) return rc;
                HIPCHK(hipEventRecord(e, st));
                HIPCHK(hipStreamWaitEvent(stream_of(wset(ci + 1)), e, 0));
            }
            if ((rc = stage_m1(x, st)) || (rc = stage_f2(x, st)) || (rc = stage_m2(x, st))) return rc;
        }
        // join: the handle's stream (pipelined: the join stream) continues only after the internal ones drained
        const hipStream_t end_st = pipe ? c->join_st : main_st;
        if (pipe) {
            HIPCHK(hipEventRecord(c->ev_joins[0], main_st));
            HIPCHK(hipStreamWaitEvent(end_st, c->ev_joins[0], 0));
        }
        for (int k = 0; k < n_work; k++) {
            const int w = wset(k);
            if (w == 0) continue; // the handle's stream: joined above (pipelined) or the end stream itself
            HIPCHK(hipEventRecord(c->ev_joins[w], c->aux[w]));
            HIPCHK(hipStreamWaitEvent(end_st, c->ev_joins[w], 0));
        }
        HIPCHK(hipEventRecord(c->ev_meta[slot], end_st));
        HIPCHK(hipEventRecord(c->ev_end, end_st));
    }
    c->timing_valid = true;
    c->last_samples = io.offsets[n] - io.offsets[0];
    if (host) {
        HIPCHK(hipStreamSynchronize(main_st));
        if (ringed) {
            HIPCHK(c->ring_up.drain());
            HIPCHK(c->ring_down.drain());
            if (full) c->ring_up.host_copy(io.results, c->pinned_res, (size_t)n * sizeof(wsx_result));
        }
    }
    return WSX_SUCCESS;
}

} // namespace

extern "C" {

// A call whose work sets do not fit the device any more -- the limit was sized from the memory that was free when the handle was
// created, and another handle, the loader's pool or the caller's own buffers have taken some since -- is planned again under a
// smaller limit instead of failing: everything of this handle is drained and released first (allocations precede the launches
// of a call: nothing of the failed attempt is in flight), then the limit becomes half of what it was or 60 % of what is free
// now, whichever is less.  Same results: the chunk plan never shows in them.
static int run_batch_retry(wsx_caller *c, const BatchIO &io, bool full)
{
    for (int attempt = 0;; attempt++) {
        g_alloc_oom = false;
        (void)hipGetLastError(); // (an error some earlier call of this thread left behind -- the host program's own failed allocation,
                                 // say -- is not this call's: the checks behind the launches ask for the last error)
        const int rc = run_batch(c, io, full);
        if (rc == WSX_SUCCESS || !g_alloc_oom || attempt >= 6 || c->ws_limit <= (256ull << 20)) return rc;
        (void)hipDeviceSynchronize();
        for (auto &w : c->work)
            for (DeviceBuf *b : {&w.samples, &w.reads, &w.bp, &w.stage_sig, &w.stage_out, &w.reps, &w.smooth}) b->release();
        size_t free_b = 0, total_b = 0;
        uint64_t next = c->ws_limit / 2;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) next = std::min<uint64_t>(next, (uint64_t)((double)free_b * 0.6));
        c->ws_limit = std::max<uint64_t>(next, 256ull << 20);
    }
}

int wsx_call_batch(wsx_caller *c, int mem, const double *signal, const int64_t *offsets, const int32_t *automaton_id,
                   int64_t n_reads, wsx_result *results, const wsx_traces *traces)
try {
    BatchIO io{};
    io.mem = mem;
    io.signal = signal;
    io.offsets = offsets;
    io.aut_id = automaton_id;
    io.n = n_reads;
    io.results = results;
    if (traces) io.traces = *traces;
    return run_batch_retry(c, io, true);
} catch (...) {
    return wsx_internal_on_exception();
}

int wsx_call_batch_reads(wsx_caller *c, const double *const *reads, const int64_t *lengths, const int32_t *automaton_id,
                         int64_t n_reads, wsx_result *results, const wsx_traces *traces)
try {
    if (!reads || !lengths || n_reads < 0) {
        g_err = "wsx_call_batch_reads: null argument";
        return WSX_ERR_INVALID;
    }
    std::vector<int64_t> offsets((size_t)n_reads + 1, 0);
    for (int64_t r = 0; r < n_reads; r++) {
        if (lengths[r] < 0 || (lengths[r] > 0 && !reads[r])) {
            g_err = "wsx_call_batch_reads: negative length or null read";
            return WSX_ERR_INVALID;
        }
        offsets[r + 1] = offsets[r] + lengths[r];
    }
    BatchIO io{};
    io.mem = WSX_MEM_HOST;
    io.signal = nullptr;
    io.read_ptrs = reads;
    io.offsets = offsets.data();
    io.aut_id = automaton_id;
    io.n = n_reads;
    io.results = results;
    if (traces) io.traces = *traces;
    return run_batch_retry(c, io, true);
} catch (...) {
    return wsx_internal_on_exception();
}

int wsx_warp_batch(wsx_caller *c, int mem, const double *signal, const int64_t *offsets, const int32_t *automaton_id,
                   int64_t n_reads, const uint8_t *mask, uint16_t *trace, double *end_cost, double *last_row,
                   int32_t last_row_stride, int32_t *status)
try {
    BatchIO io{};
    io.mem = mem;
    io.signal = signal;
    io.offsets = offsets;
    io.aut_id = automaton_id;
    io.n = n_reads;
    io.mask = mask;
    io.trace = trace;
    io.end_cost = end_cost;
    io.last_row = last_row;
    io.last_row_stride = last_row_stride;
    io.status = status;
    if (last_row && c && last_row_stride < c->max_states) {
        g_err = "last_row_stride smaller than the largest automaton";
        return WSX_ERR_INVALID;
    }
    return run_batch_retry(c, io, false);
} catch (...) {
    return wsx_internal_on_exception();
}

int wsx_caller_last_timing(wsx_caller *c, double *dp_kernel_ms, int32_t *dp_launches, double *total_ms)
{
    if (!c) return WSX_ERR_INVALID;
    if (!c->timing_valid) {
        g_err = "no completed batch to time";
        return WSX_ERR_INVALID;
    }
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipEventSynchronize(c->ev_end));
    float tot = 0.f;
    HIPCHK(hipEventElapsedTime(&tot, (c->timing_window && c->window_open) ? c->ev_window : c->ev_begin, c->ev_end));
    double dp = 0.0;
    for (size_t i = 0; i < c->dp_events_used; i++) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, c->dp_events[i].first, c->dp_events[i].second));
        dp += ms;
    }
    if (dp_kernel_ms) *dp_kernel_ms = dp;
    if (dp_launches) *dp_launches = (int32_t)c->dp_events_used;
    if (total_ms) *total_ms = tot;
    return WSX_SUCCESS;
}

int wsx_caller_workspace(wsx_caller *c, uint64_t *bytes_allocated, double *bytes_per_sample)
{
    if (!c) return WSX_ERR_INVALID;
    uint64_t total = c->aut_blob.cap + c->aut_table.cap;
    for (const auto &b : c->aut_retired) total += b.cap;
    for (auto &b : c->meta) total += b.cap;
    for (auto &w : c->work)
        for (const DeviceBuf *b : {&w.samples, &w.reads, &w.bp, &w.stage_sig, &w.stage_out, &w.reps, &w.smooth}) total += b->cap;
    // (per sample of the most recent call: the caller's own buffers; the loader's pool -- wsx_prepare_signals, sized by its
    // own calls -- counts towards the total only)
    if (bytes_per_sample) *bytes_per_sample = c->last_samples > 0 ? (double)total / (double)c->last_samples : 0.0;
    for (auto &b : c->prep_pool) total += b.cap;
    if (bytes_allocated) *bytes_allocated = total;
    return WSX_SUCCESS;
}

int wsx_caller_fill_intervals(wsx_caller *c, double *begin_ms, double *end_ms, int32_t *reads, int32_t capacity,
                              int32_t *n_out)
{
    if (!c || !n_out || capacity < 0) return WSX_ERR_INVALID;
    if (!c->timing_valid) {
        g_err = "no completed batch to time";
        return WSX_ERR_INVALID;
    }
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipEventSynchronize(c->ev_end));
    const hipEvent_t origin = (c->timing_window && c->window_open) ? c->ev_window : c->ev_begin;
    *n_out = (int32_t)c->dp_events_used;
    for (size_t i = 0; i < c->dp_events_used && (int32_t)i < capacity; i++) {
        float b = 0.f, e = 0.f;
        HIPCHK(hipEventElapsedTime(&b, origin, c->dp_events[i].first));
        HIPCHK(hipEventElapsedTime(&e, origin, c->dp_events[i].second));
        if (begin_ms) begin_ms[i] = b;
        if (end_ms) end_ms[i] = e;
        if (reads) reads[i] = c->dp_reads[i];
    }
    return WSX_SUCCESS;
}

} // extern "C"
