// ctx.h -- host-side context of libw2rap_step2.so and the launcher prototypes of the
// phase files.  One context drives one GPU on one HIP stream.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <cstdlib>
#include <string>
#include <utility>
#include <vector>
#include "../../include/w2rap_step2.h"
#include "common.h"

namespace w2 {

struct Ctx;
}  // namespace w2
struct w2rap_step2_ctx;
namespace w2 {
struct Ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;      // side stream: the dictionary build runs here while the counting kernel is still busy
    unsigned long long* h_pinned = nullptr;   // 64 pinned words for asynchronous counter read-back
    std::string err;
    int sm_count = 256;

    // ---- reads (device) ----
    uint64_t n = 0;
    const uint8_t* d_bases = nullptr;
    const uint64_t* d_boff = nullptr;
    const uint32_t* d_len = nullptr;
    const uint8_t* d_quals = nullptr;
    const uint64_t* d_qoff = nullptr;
    uint32_t max_len = 0;
    std::vector<void*> owned_reads;     // freed when reads are replaced / ctx destroyed

    // ---- a1 ----
    uint16_t* d_good = nullptr;
    uint32_t min_qual = 7, min_freq = 4;

    // ---- a2-a5 ----
    uint64_t M = 0, D = 0, S = 0;
    uint64_t hist[101] = {0};
    uint32_t NB = 0;                    // buckets
    uint32_t* d_bcount = nullptr;       // [NB] records per bucket
    unsigned long long part_kmers[64] = {0};   // k-mer instances per owner rank (multi-GPU partition only)
    uint64_t* d_bbase = nullptr;        // [NB+1] first record of each bucket
    uint32_t* d_recs = nullptr;         // records, REC_DWORDS each
    uint64_t nrec = 0;
    uint64_t solid_cap = 0;
    uint64_t* d_shi = nullptr;          // solid k-mers (device order)
    uint64_t* d_slo = nullptr;
    uint32_t* d_scc = nullptr;          // count | ctx << 8
    Slot* d_table = nullptr;
    uint64_t tcap = 0;
    unsigned long long* d_filter32 = nullptr;   // absence filter over the 31-mers of the unipath sequences (built with the graph); f32words-1 = mask
    uint64_t f32words = 0;
    uint8_t* d_sctx = nullptr;          // [S] pruned context
    void* d_nbr = nullptr;              // [2S] the single successor / predecessor of each k-mer as an oriented node (k_prune): u32, or u64 with wide_ids
    bool wide_ids = false;              // node ids are 64-bit words (S >= 2^31 - 1, or forced)
    uint64_t* d_chunk_start = nullptr;  // K3's emits: first solid k-mer of each (bucket, class) chunk ...
    uint32_t* d_chunk_cnt = nullptr;    // ... and their number (single-GPU path only; the bucket-local prune works on them)
    uint64_t nchunks = 0;
    KRec* d_srec = nullptr;             // [S] {hi, lo, KDef}: x = unipath id | (lies on it reverse-complemented) << 31, y = offset,
                                        //     z | (w & 0xFF) << 32 = first base of the unipath in the edge stream, w >> 8 = its k-mers
                                        //     (read pathing through the dictionary: one GPU, use_index == false)
    bool use_index = false;             // read pathing asks the index below instead of d_table + d_srec (sharded dictionary; W2RAP_PATH_INDEX=1)
    uint4* d_index = nullptr;           // minimizer-sampled index over the edge stream (common.h EdgeIndex)
    uint64_t index_cap = 0, index_entries = 0;
    uint4* d_xindex = nullptr; uint64_t xindex_cap = 0, xindex_kmers = 0;      // the exact table beside it (common.h), 0 slots: none
    bool index_prebuilt = false, bits_ready = false, filter_prebuilt = false;      // (sharded graph phase: the index / the packed stream exist before graph_finish)
    // ---- sliced counting (multi-GPU: slice k's solid k-mers are exchanged while slice k+1 is counted)
    unsigned cs_ns = 0;                 // slices launched so far
    unsigned cs_planned = 0;            // slices of the pending count (0: none pending)
    uint32_t cs_nbl = 0, cs_nseg = 0; const uint32_t* cs_recs = nullptr;
    bool cs_short_first = false;        // slice 0 covers half as many buckets as the others
    hipEvent_t cs_ev[16] = {};
    unsigned long long* cs_cnt = nullptr;   // device counters of the pending count
    uint64_t* cs_off = nullptr;
    uint32_t cs_chunk_cap = 0;
    uint8_t* d_unres = nullptr;         // [solid_cap] context bits the chunk-local prune left open (fused into k_count_fp's emit: W2RAP_FUSED_PRUNE)
    bool fused_prune = false;           // this count's k_count_fp launches do the chunk-local prune; k_prune_local only sees the chunks listed below
    std::vector<std::pair<uint64_t, uint64_t>> unfused_chunks;   // [first, end) chunk numbers written by the list kernel (not pruned yet)
    uint32_t* cs_defer = nullptr;       // [2 + buckets of the count] k_count_fp's deferred buckets: [0] their number, [1] the list kernel's queue
    // ---- dictionary under construction (dict_begin / dict_append / dict_end): gathered solid k-mers, inserted on the side stream
    uint64_t* g_hi = nullptr; uint64_t* g_lo = nullptr; uint32_t* g_cc = nullptr;
    uint64_t* g_cstart = nullptr; uint32_t* g_ccnt = nullptr;
    uint64_t g_cap = 0, g_n = 0, g_ccap = 0, g_nc = 0;
    bool g_open = false;
    hipEvent_t g_copied = nullptr;      // recorded on the side stream behind the copies of the latest dict_append
    bool quality_done = false, counted = false, graphed = false, pathed_done = false;
    bool table_built = false;           // d_table already filled (overlapped with counting)

    // ---- a7 ----
    uint64_t E = 0;                     // unipaths
    uint64_t rank_ends = 0;             // chain ends seen by the last list ranking (bounds the head list)
    uint32_t* d_edge_nk = nullptr;      // [E] k-mers per edge
    uint64_t* d_edge_off = nullptr;     // [E+1] base offset into d_edge_codes
    uint8_t* d_edge_codes = nullptr;    // unpacked bases of all edges, canonical orientation
    uint8_t* d_edge_bits = nullptr;     // the same bases as ONE 2-bit LSB-first stream (base g at bits 2g), +16 B slack
    uint64_t edge_bases = 0;
    // ---- a8 ----
    uint64_t NO = 0, NV = 0;            // edge objects, vertices
    int32_t* d_fwdX = nullptr;          // [E]
    int32_t* d_revX = nullptr;          // [E]
    uint32_t* d_obj_edge = nullptr;     // [NO] edge<<1 | rc
    ObjRec* d_otab = nullptr;           // [NO] successors by next base + place in the edge stream (read pathing's walk across unipath ends)
    int32_t* d_left = nullptr;          // [NO]
    int32_t* d_right = nullptr;         // [NO]
    uint64_t* d_from_off = nullptr;     // [NV+1]
    int32_t* d_from_v = nullptr;        // [NO]
    int32_t* d_from_e = nullptr;
    uint64_t* d_to_off = nullptr;
    int32_t* d_to_v = nullptr;
    int32_t* d_to_e = nullptr;
    // ---- a9-a12 ----
    int32_t* d_path_offset = nullptr;   // [n]
    uint64_t* d_path_off = nullptr;     // [n+1]
    int32_t* d_path_edges = nullptr;
    uint64_t path_total = 0;
    uint64_t n_pathed = 0, n_multipathed = 0;
    float ms_count = 0, ms_graph = 0, ms_path = 0;

    uint64_t ld_done = 0;               // solid k-mers already inserted into the owner's own dictionary under the counting (local_dict_slice)
    void* shard = nullptr;              // state of the sharded graph phase (step2_shard.hip), between shard_begin and the next count
    std::vector<void*> owned;           // everything else
    void* pump = nullptr;               // pinned staging ring for host <-> device copies of big arrays (step2_run.hip), created on first use
    // ---- qualities that arrive LATE (w2rap_step2_run on host arrays): the quality windows (K0) run on a one-bit-per-base mask made on the
    // host side of the pump, while the raw bytes -- which only the scores of the path extension read -- travel on a copy stream under the
    // counting and the graph phase
    const uint32_t* d_qmask = nullptr;  // bit i = quality i >= qmask_min_qual (one array over all reads, as d_quals)
    int qmask_min_qual = -1;
    int hint_min_qual = -1;             // set around set_reads by w2rap_step2_run: the threshold to make the mask for (-1: none, qualities travel first)
    bool hint_graph_only = false;       // ... and: nobody will path reads (W2RAP_F_GRAPH_ONLY), the raw qualities need not travel at all
    bool quals_absent = false;          // the raw qualities were never uploaded (graph-only call): read pathing refuses
    uint8_t* d_qring = nullptr;         // the late upload's 6-bit-packed qualities, until c.stream has unpacked them (step2_run.hip)
    void* pump2 = nullptr;              // the background upload's own staging ring ...
    hipStream_t copy_stream = nullptr;  // ... and stream
    void* quals_job = nullptr;          // the pending background upload (step2_run.hip), nullptr when there is none
    void* mask_job = nullptr; void* h_mask = nullptr; size_t h_mask_bytes = 0;       // the mask being made, and the host buffer it is made in (kept between calls)
    uint32_t n_passes = 0;              // count_kmers: hash-range passes of the counting phase (0 = choose from free HBM)
    unsigned pass = 0, npass = 1;       // the pass being counted / their number (phase_count)
    unsigned long long* pass_cnt = nullptr;   // device counters carried from one pass to the next

    // ---- per-kernel timing (hipEvents on c.stream), summed per kernel name
    struct ProfEv { const char* name; hipEvent_t a, b; };
    struct ProfSum { std::string name; double ms = 0; uint64_t launches = 0; };
    std::vector<ProfEv> prof_pending;
    std::vector<ProfSum> prof_sums;
    bool profiling = true;
    void pbegin(const char* name, hipStream_t st = nullptr) {
        if (!profiling) return;
        ProfEv e{name, nullptr, nullptr};
        (void)hipEventCreate(&e.a); (void)hipEventCreate(&e.b);
        (void)hipEventRecord(e.a, st ? st : stream);
        prof_pending.push_back(e);
    }
    void pend(hipStream_t st = nullptr) {
        if (!profiling || prof_pending.empty()) return;
        (void)hipEventRecord(prof_pending.back().b, st ? st : stream);
    }
    void presolve() {                    // call after a stream synchronize
        for (auto& e : prof_pending) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
                ProfSum* s = nullptr;
                for (auto& x : prof_sums) if (x.name == e.name) { s = &x; break; }
                if (!s) { prof_sums.push_back(ProfSum{e.name, 0, 0}); s = &prof_sums.back(); }
                s->ms += ms; s->launches += 1;
            }
            (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b);
        }
        prof_pending.clear();
    }

    // Device memory is recycled inside the context: a released block is parked by byte size and
    // handed out again for the next request of that size (the phases of consecutive runs ask for
    // identical sizes), which removes hipMalloc/hipFree -- page-table work on tens of GB -- from
    // the steady state.  trim() returns the parked blocks to the driver.
    std::vector<std::pair<size_t, void*>> parked;
    std::vector<std::pair<void*, size_t>> sizes;       // live blocks handed out by alloc()
    int mem_trace = -1; uint64_t traced_peak = 0;     // W2RAP_TRACE_MEM=1: the live blocks at every new peak, on stderr
    uint64_t live_bytes = 0, peak_bytes = 0;            // bytes of the live blocks now / their maximum since the last reset (w2rap_step2_device_peak_bytes)
    // Size classes, so that a request a few per cent off an earlier one -- list lengths that differ from run to run -- still finds its parked
    // block: 1/16..1/32 of the request's magnitude below 256 MiB (up to ~6 % more), 1/128..1/256 from there on (under 1 %: the multi-GB arrays
    // of a large job -- solid arrays, record buffers, link arrays -- are what the memory planning prices)
    static size_t size_class(size_t bytes) {
        size_t g = 256;
        if (bytes < (256u << 20)) { while ((g << 5) <= bytes) g <<= 1; }
        else { g = 1u << 20; while ((g << 8) <= bytes) g <<= 1; }
        return (bytes + g - 1) & ~(g - 1);
    }
    void trim() {
        for (auto& b : parked) (void)hipFree(b.second);
        parked.clear();
    }
    template <class T>
    T* alloc(size_t count, bool track = true) {
        void* p = nullptr;
        size_t bytes = (count ? count : 1) * sizeof(T);
        bytes = size_class(bytes);
        for (size_t i = 0; i < parked.size(); ++i)
            if (parked[i].first == bytes) { p = parked[i].second; parked[i] = parked.back(); parked.pop_back(); break; }
        if (!p) {
            hipError_t e = hipMalloc(&p, bytes);
            if (e != hipSuccess) { (void)hipGetLastError(); trim(); e = hipMalloc(&p, bytes); }
            // still short: the idle contexts of the process-wide cache (step2_run.hip) park tens of GB each in their own pools
            if (e != hipSuccess) { (void)hipGetLastError(); if (w2rap_step2_trim_cached() > 0) { (void)hipSetDevice(device); e = hipMalloc(&p, bytes); } }
            if (e != hipSuccess) {
                err = std::string("hipMalloc(") + std::to_string(bytes) + " B): " + hipGetErrorString(e);
                return nullptr;
            }
        }
        sizes.emplace_back(p, bytes);
        live_bytes += bytes;
        if (live_bytes > peak_bytes) {
            peak_bytes = live_bytes;
            if (mem_trace < 0) mem_trace = getenv("W2RAP_TRACE_MEM") ? 1 : 0;
            if (mem_trace && live_bytes > traced_peak + (1ull << 30)) {      // the live blocks of 64 MiB and more at every new peak (a GiB apart)
                traced_peak = live_bytes;
                std::string l;
                for (auto& x : sizes) if (x.second >= (64u << 20)) l += " " + std::to_string(x.second >> 20);
                fprintf(stderr, "[w2rap] device memory: new peak %.2f GB live; blocks (MiB):%s\n", live_bytes / 1e9, l.c_str());
            }
        }
        if (track) owned.push_back(p);
        return (T*)p;
    }
    void park(void* p) {
        for (size_t i = 0; i < sizes.size(); ++i)
            if (sizes[i].first == p) { live_bytes -= sizes[i].second; parked.emplace_back(sizes[i].second, p); sizes[i] = sizes.back(); sizes.pop_back(); return; }
        (void)hipFree(p);
    }
    void release(void* p) {
        if (!p) return;
        for (size_t i = 0; i < owned.size(); ++i)
            if (owned[i] == p) { owned[i] = owned.back(); owned.pop_back(); break; }
        park(p);
    }
    void free_all() {
        for (void* p : owned) park(p);
        owned.clear();
    }
};

// Test switches read from the environment never act silently: a set hook is announced on stderr every time it is honoured.
// Hooks that make RESULTS wrong (a removed stream dependency, ablated pathing) exist only in builds with -DW2RAP_TESTING.
inline bool test_hook(const char* name) {
    if (!getenv(name)) return false;
    fprintf(stderr, "[w2rap] WARNING: test hook %s is set -- this is not a production configuration\n", name);
    return true;
}

#define W2_HIP(call)                                                                         \
    do {                                                                                     \
        hipError_t e__ = (call);                                                             \
        if (e__ != hipSuccess) {                                                             \
            c.err = std::string(#call) + ": " + hipGetErrorString(e__) + " (" + __FILE__ + ":" + \
                    std::to_string(__LINE__) + ")";                                          \
            return W2RAP_E_HIP;                                                              \
        }                                                                                    \
    } while (0)
#define W2_ALLOC(ptr, T, count)                          \
    do {                                                 \
        ptr = c.alloc<T>(count);                         \
        if (!ptr) return W2RAP_E_HIP;                    \
    } while (0)
#define W2_TRY(expr)                   \
    do {                               \
        int rc__ = (expr);             \
        if (rc__) return rc__;         \
    } while (0)

// timed kernel launch: LAUNCH(c, "name", kernel, grid, block, lds, args...)
#define LAUNCH(c, name, kern, grid, block, lds, ...)                          \
    do {                                                                      \
        (c).pbegin(name);                                                     \
        hipLaunchKernelGGL(kern, grid, block, lds, (c).stream, __VA_ARGS__);  \
        (c).pend();                                                           \
    } while (0)

// ---- small host-side helpers shared by the step drivers
// elapsed device time on a stream, between construction and stop()
struct Timer {
    hipEvent_t a = nullptr, b = nullptr; hipStream_t st;
    explicit Timer(hipStream_t s) : st(s) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventRecord(a, st); }
    float stop() { float ms = 0; (void)hipEventRecord(b, st); (void)hipEventSynchronize(b); (void)hipEventElapsedTime(&ms, a, b); return ms; }
    ~Timer() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); }
    Timer(const Timer&) = delete; Timer& operator=(const Timer&) = delete;
};
// a result array -> freshly malloc'ed host memory (small ones: copy queued on the context's stream, synchronise before reading; big
// ones go through the pinned staging pump and are complete on return)
// host memory for a result array (the caller frees it with free()): big arrays on 2-MB boundaries with transparent huge pages asked for --
// a fresh malloc'ed GB takes a page fault per 4 KB at its first touch, which was most of the 59 ms the 0.9 GB of read paths took to come
// down (round 5 trace: 15.7 GB/s); with huge pages the first touch is 512x rarer
void* host_result_alloc(size_t bytes);                                  // step2_run.hip
template <class T>
inline int dl(Ctx& c, T** host, const T* dev, uint64_t n) {
    *host = (T*)host_result_alloc((n ? n : 1) * sizeof(T));
    if (!*host) { c.err = "out of host memory"; return W2RAP_E_HIP; }
    if (n * sizeof(T) >= (8u << 20)) return pump_download(c, *host, dev, n * sizeof(T));
    if (n) W2_HIP(hipMemcpyAsync(*host, dev, n * sizeof(T), hipMemcpyDeviceToHost, c.stream));
    return 0;
}
// a host array -> a block of the context's pool, `pad` + 1 zeroed elements behind it
template <class T>
inline int up_pooled(Ctx& c, T** dev, const T* host, uint64_t n, uint64_t pad = 0) {
    T* p = c.alloc<T>(n + pad + 1);
    if (!p) return W2RAP_E_HIP;
    W2_HIP(hipMemsetAsync(p + n, 0, (pad + 1) * sizeof(T), c.stream));
    if (n && !host) { c.err = "null host array"; return W2RAP_E_ARG; }
    if (host) { if (n) W2_HIP(hipMemcpyAsync(p, host, n * sizeof(T), hipMemcpyHostToDevice, c.stream)); }
    else W2_HIP(hipMemsetAsync(p, 0, sizeof(T), c.stream));            // an empty input given as a null pointer: an explicit zero (first offset)
    *dev = p;
    return 0;
}

// the same on an explicit stream
#define LAUNCH_ON(c, st, name, kern, grid, block, lds, ...)                   \
    do {                                                                      \
        (c).pbegin(name, st);                                                 \
        hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__);          \
        (c).pend(st);                                                         \
    } while (0)

// host <-> device copies through pinned staging buffers filled / drained by worker threads, and the workers themselves (step2_run.hip)
int device_copy_async(Ctx& c, void* dst, const void* src, uint64_t nbytes, hipStream_t st);   // step2_capi.hip: the library's copy kernel (same device)
int pump_upload(Ctx& c, void* d, const void* h, size_t bytes);        // queued on c.stream; the host array may be reused when it returns
int pump_download(Ctx& c, void* h, const void* d, size_t bytes);      // complete when it returns
void pump_free(Ctx& c);
int quality_mask_begin(Ctx& c, const uint8_t* h_quals, uint64_t nq, uint32_t min_qual);   // threads of its own make the mask in a host buffer of the context's ...
int quality_mask_upload(Ctx& c, uint32_t* d_mask);      // ... joined here; the mask goes up through the pump (queued on c.stream)
void quality_mask_cancel(Ctx& c);
int quals_upload_begin(Ctx& c, uint8_t* d_quals, const uint8_t* h_quals, uint64_t nq, uint64_t prefix_bytes, uint64_t prefix_reads);   // a host thread + the copy stream; h_quals must live until quals_wait
int quals_wait_prefix(Ctx& c, uint64_t* reads_covered);   // read pathing may start on the first prefix_reads reads: their qualities are up (0: no such prefix)
int quals_wait(Ctx& c);                 // joins the upload, makes c.stream wait for its last copy; the upload's error, if any (no-op without a pending upload)
}  // namespace w2
#include <functional>
namespace w2 {
void host_parallel_for(size_t n, const std::function<void(size_t)>& f);

// phase drivers (one per .hip file)
int phase_count(Ctx& c, uint32_t min_qual, uint32_t min_freq);          // step2_count.hip
int count_quality(Ctx& c, uint32_t min_qual);
uint32_t default_buckets(uint64_t total_kmers, uint32_t multiple_of);
int count_partition(Ctx& c, uint32_t nb, uint32_t n_parts, uint32_t pb_lo, uint32_t pb_hi);   // n_parts 0: single GPU; buckets [pb_lo, pb_hi) of nb
int count_buckets(Ctx& c, uint32_t min_freq, uint32_t nbl, uint32_t nseg, const uint32_t* d_recs, const uint32_t* d_counts, uint64_t total_kmers,
                  bool build_table = false);
int count_buckets_launch(Ctx& c, uint32_t min_freq, uint32_t nbl, uint32_t nseg, const uint32_t* d_recs, const uint32_t* d_counts, uint64_t total_kmers,
                         unsigned n_slices, bool deferred);
int count_buckets_launch_slice(Ctx& c, unsigned k);
void count_slice_bounds(const Ctx& c, unsigned k, uint32_t* b_lo, uint32_t* b_hi);
int count_buckets_slice(Ctx& c, unsigned k, uint64_t* n_solid, uint64_t* n_chunks);
int count_buckets_finish(Ctx& c);
int dict_begin(Ctx& c, uint64_t kmer_cap, uint64_t chunk_cap);
int dict_append(Ctx& c, const uint64_t* d_hi, const uint64_t* d_lo, const uint32_t* d_cc, uint64_t n, const uint64_t* d_cstart, const uint32_t* d_ccnt, uint64_t nc,
                uint64_t chunk_bias = 0);
int dict_end(Ctx& c);
void dict_abort(Ctx& c);
int count_table(Ctx& c);
int phase_graph(Ctx& c, const w2rap_edge_hint* hint);                    // step2_graph.hip
int phase_path(Ctx& c);                                                  // step2_path.hip
// the dictionary, prune and unipath phases sharded by bucket owner (step2_shard.hip): a state machine between exchanges
int local_dict_slice(Ctx& c, uint64_t n_solid, uint64_t expected_total);    // step2_count.hip: the owner's own dictionary, slice by slice on the side stream
int shard_begin(Ctx& c, unsigned rank, unsigned world, const uint64_t* solid_per_rank, uint32_t n_buckets, uint32_t n_passes, const w2rap_edge_hint* hint);
int shard_next(Ctx& c, w2rap_xchg* x);
int shard_recv(Ctx& c, const uint64_t* recv_count, uint32_t elem_bytes, void** d_recv);
int shard_host_words(Ctx& c, const uint64_t* words);
int shard_info(Ctx& c, uint64_t out[8]);
void shard_free(Ctx& c);
int index_hard_slice(Ctx& c, const uint4* d_all, uint64_t n_all, unsigned rank, unsigned world, uint4** d_hard, uint64_t* n_hard);   // sharded: this rank's part of the entry list
int index_hard_apply(Ctx& c, const uint4* d_hard, uint64_t n_hard);     // ... and the gathered hard entries: marks + the exact table
int index_harden(Ctx& c);                                               // step2_graph.hip: marks the index keys with many entries, builds the exact table of their k-mers
int build_index(Ctx& c);                                                 // step2_graph.hip: the pathing index over c.d_edge_bits
int index_entries_slice(Ctx& c, unsigned rank, unsigned world, uint4** d_list, uint64_t* n_list);   // this rank's share of the index entries, as a list
int index_from_entries(Ctx& c, const uint4* d_all, uint64_t n_all, bool harden = true);                                  // the table from every rank's list
uint64_t filter32_words(const Ctx& c);
int filter32_slice(Ctx& c, unsigned rank, unsigned world, unsigned long long** d_slice, uint64_t* n_words);   // this rank's words of the absence filter
EdgeIndex edge_index(const Ctx& c);
int index_probe_all(Ctx& c, int32_t* d_edge, uint32_t* d_off);           // (edge, offset) of every solid k-mer through the index
// list ranking over N oriented nodes linked by nxt0 (step2_graph.hip): nxt = the chain end every node reaches, rnk = its distance,
// cyc = lies on a circle; shi/slo (60-mers, may be null) only feed the middle-base output `mid`
int run_ranking(Ctx& c, uint64_t N, const uint32_t* nxt0, uint32_t* nxt, uint32_t* rnk, unsigned long long* w, uint8_t* cyc, uint8_t* mid,
                uint32_t* d_flags, const uint64_t* shi, const uint64_t* slo, bool use_chunks);
int decode_pq(Ctx& c, const uint8_t* d_pq, const uint64_t* d_pqoff, uint8_t* d_quals, const uint64_t* d_qoff);  // step2_count.hip

// device-wide primitives (step2_prims.hip; rocPRIM underneath)
int sort_pairs_u64(Ctx& c, uint64_t* keys, uint32_t* vals, uint64_t n, int begin_bit, int end_bit);   // stable, in place
int exclusive_scan_u32_to_u64(Ctx& c, const uint32_t* in, uint64_t* out, uint64_t n);                  // out[n] = total
int exclusive_scan_packed_bytes(Ctx& c, const uint32_t* len, uint64_t* out, uint64_t n);           // out[i] = sum of ceil(len[j] / 4) over j < i
int exclusive_scan_u64(Ctx& c, const uint64_t* in, uint64_t* out, uint64_t n);                         // out[n] = total
int exclusive_scan_is_self(Ctx& c, const uint32_t* a, uint64_t* out, uint64_t n);                      // of the flags a[i] == i
int max_u32(Ctx& c, const uint32_t* in, uint64_t n, uint32_t* result);
int inclusive_max_scan_u32(Ctx& c, const uint32_t* in, uint32_t* out, uint64_t n);                       // out[i] = max(in[0..i])

}  // namespace w2

struct w2rap_step2_ctx { w2::Ctx c; };
// contexts from the process-wide cache (step2_run.hip): the one-shot entry points of Steps 1-3 and the GFA dump take theirs here, so
// that a second call in one process finds the pool of device blocks of the first
extern "C" w2rap_step2_ctx* w2rap_step2_acquire(int device, char* err, size_t errlen);
extern "C" void w2rap_step2_release(w2rap_step2_ctx*);
