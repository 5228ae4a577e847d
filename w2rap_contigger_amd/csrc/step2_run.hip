// step2_run.hip -- the one-shot boundary of libw2rap_step2.so: w2rap_step2_run on HOST buffers, on one or several GPUs.
//
//   buildReadQGraph(...) + FixPaths(...)   modules/w2rap-contigger.cc:338-340, paths/long/BuildReadQGraph.h:24-29
//
// Host-side plumbing only (no CPU implementation of any phase):
//  * a process-wide cache of idle contexts: a second call in the same process (and Steps 1, 3 and the GFA dump, which take their
//    context here too) finds its streams and, above all, its pool of device blocks again -- the page-table work of tens of GB of
//    hipMalloc / hipFree is paid once per process, not once per call;
//  * host <-> device copies of the big arrays through a ring of pinned staging buffers filled / drained by a few worker threads
//    (a pageable hipMemcpy moves ~25 GB/s, this pump ~2x that): the transfer of the reads is what a one-shot call costs;
//  * n_gpus > 1: one host thread and one context per device.  Reads are sharded by contiguous ranges of whole pairs; every k-mer
//    bucket has one owner (equal contiguous bucket ranges); the super-k-mer records travel by direct peer copies over
//    xGMI -- an all-to-all written as (world-1) device-to-device copies per rank, all links at once --, every owner counts its buckets,
//    the solid k-mers are gathered by every rank in owner order (so that all ranks number the k-mers identically), each rank
//    builds the (replicated) graph and paths its own reads.  Template: the in-process MAP -> SWIZZLE -> REDUCE of
//    MapReduceEngine.h:320-361.  The Python host code (w2rap_contigger_amd/dist.py, one PROCESS per GPU over RCCL) is the variant
//    that overlaps the exchange with the counting; this one serves a C++ caller that makes one in-process call.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <sys/mman.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include "ctx.h"

using namespace w2;


namespace w2 {

// ------------------------------------------------------------------------------------------------ worker threads
// A small persistent pool: parallel_for(n, f) runs f(0..n-1) on the workers and the caller.  Used for host memcpy into / out of the
// pinned staging buffers and for the validation sweeps over the callers' offset arrays.
class HostPool {
public:
    static HostPool& get() { static HostPool p(0); return p; }
    // a second pool of the same size for the one job that must not queue behind the copies of the first: packing the late qualities to 6 bits.
    // (Measured on the 256-core host, 7.5 GB: 16 threads pack in 60 ms -- the wire then takes 110 ms for the packed bytes --, 40 threads
    // in 90-140 ms, 96 in 280-330 ms: more threads only fight over the host's memory system.)
    static HostPool& wide() { static HostPool p(std::thread::hardware_concurrency() >= 64 ? 16u : 0u); return p; }
    unsigned size() const { return (unsigned)workers_.size() + 1; }
    void parallel_for(size_t n, const std::function<void(size_t)>& f) {
        if (n <= 1 || workers_.empty()) { for (size_t i = 0; i < n; ++i) f(i); return; }
        std::unique_lock<std::mutex> run(run_mu_);                // one parallel_for at a time
        {
            std::lock_guard<std::mutex> g(mu_);
            fn_ = &f; n_ = n; next_ = 0; left_ = n; ++gen_;
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> g(mu_);
        done_.wait(g, [&] { return left_ == 0; });
        fn_ = nullptr;
    }
private:
    explicit HostPool(unsigned want) {
        unsigned hw = std::thread::hardware_concurrency();
        unsigned n = hw >= 64 ? 16 : hw > 16 ? 8 : hw > 2 ? hw / 2 : 1;      // (16 on a big host: the quality mask of the late-quality upload is a pass over 7.5 GB of host memory)
        if (want) n = want;
        if (const char* v = getenv(want ? "W2RAP_PACK_THREADS" : "W2RAP_HOST_THREADS")) n = (unsigned)std::max(1, atoi(v));
        for (unsigned i = 1; i < n; ++i) workers_.emplace_back([this] { loop(); });
    }
    ~HostPool() {
        { std::lock_guard<std::mutex> g(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    void work() {
        for (;;) {
            size_t i;
            const std::function<void(size_t)>* f;
            {
                std::lock_guard<std::mutex> g(mu_);
                if (!fn_ || next_ >= n_) return;
                i = next_++; f = fn_;
            }
            (*f)(i);
            std::lock_guard<std::mutex> g(mu_);
            if (--left_ == 0) done_.notify_all();
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
            }
            work();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex mu_, run_mu_;
    std::condition_variable cv_, done_;
    const std::function<void(size_t)>* fn_ = nullptr;
    size_t n_ = 0, next_ = 0, left_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};

void host_parallel_for(size_t n, const std::function<void(size_t)>& f) { HostPool::get().parallel_for(n, f); }
static void host_parallel_for_wide(size_t n, const std::function<void(size_t)>& f) { HostPool::wide().parallel_for(n, f); }

static void parallel_memcpy(void* dst, const void* src, size_t bytes) {
    const size_t piece = 4u << 20;
    if (bytes <= piece) { std::memcpy(dst, src, bytes); return; }
    const size_t n = (bytes + piece - 1) / piece;
    host_parallel_for(n, [&](size_t i) {
        const size_t a = i * piece, b = std::min(bytes, a + piece);
        std::memcpy((uint8_t*)dst + a, (const uint8_t*)src + a, b - a);
    });
}

void* host_result_alloc(size_t bytes) {
    constexpr size_t HUGE = 2u << 20;
    if (bytes < (8u << 20) || getenv("W2RAP_NO_HUGEPAGES")) return std::malloc(bytes);
    const size_t len = (bytes + HUGE - 1) & ~(HUGE - 1);
    void* p = std::aligned_alloc(HUGE, len);
    if (p) (void)madvise(p, len, MADV_HUGEPAGE);
    return p;
}

// ------------------------------------------------------------------------------------------------ pinned staging pump
struct Pump {
    static constexpr size_t SLOT = 64u << 20;
    static constexpr int NSLOT = 4;
    uint8_t* slot[NSLOT] = {};
    hipEvent_t ev[NSLOT] = {};
    bool used[NSLOT] = {};
    bool ok = false;
    int init() {
        for (int i = 0; i < NSLOT; ++i) {
            if (hipHostMalloc((void**)&slot[i], SLOT, hipHostMallocDefault) != hipSuccess) return 1;
            if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) return 1;
        }
        ok = true;
        return 0;
    }
    void destroy() {
        for (int i = 0; i < NSLOT; ++i) { if (slot[i]) (void)hipHostFree(slot[i]); if (ev[i]) (void)hipEventDestroy(ev[i]); slot[i] = nullptr; ev[i] = nullptr; }
        ok = false;
    }
};

static Pump* pump_of(Ctx& c) {
    if (!c.pump) {
        Pump* p = new Pump;
        if (p->init()) { p->destroy(); delete p; return nullptr; }
        c.pump = p;
    }
    return static_cast<Pump*>(c.pump);
}
void pump_free(Ctx& c) {
    (void)quals_wait(c);
    quality_mask_cancel(c);
    if (c.h_mask) { std::free(c.h_mask); c.h_mask = nullptr; c.h_mask_bytes = 0; }
    if (c.pump) { static_cast<Pump*>(c.pump)->destroy(); delete static_cast<Pump*>(c.pump); c.pump = nullptr; }
    if (c.copy_stream) { (void)hipStreamSynchronize(c.copy_stream); }
    if (c.pump2) { static_cast<Pump*>(c.pump2)->destroy(); delete static_cast<Pump*>(c.pump2); c.pump2 = nullptr; }
    if (c.d_qring) { c.park(c.d_qring); c.d_qring = nullptr; }
    if (c.copy_stream) { (void)hipStreamDestroy(c.copy_stream); c.copy_stream = nullptr; }
}

// host (pageable) -> device: the host side of piece i+1 is copied into its pinned slot while piece i travels
int pump_upload(Ctx& c, void* d, const void* h, size_t bytes) {
    if (!bytes) return 0;
    Pump* p = bytes >= (8u << 20) && !getenv("W2RAP_NO_PUMP") ? pump_of(c) : nullptr;
    if (!p) { W2_HIP(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c.stream)); return 0; }
    size_t off = 0; int k = 0;
    while (off < bytes) {
        const int s = k % Pump::NSLOT;
        const size_t n = std::min(Pump::SLOT, bytes - off);
        if (p->used[s]) W2_HIP(hipEventSynchronize(p->ev[s]));
        parallel_memcpy(p->slot[s], (const uint8_t*)h + off, n);
        W2_HIP(hipMemcpyAsync((uint8_t*)d + off, p->slot[s], n, hipMemcpyHostToDevice, c.stream));
        W2_HIP(hipEventRecord(p->ev[s], c.stream));
        p->used[s] = true;
        off += n; ++k;
    }
    return 0;
}
// the same with the pieces PRODUCED into the pinned slots (fill(dst, off, n): bytes [off, off + n) of what the device array shall hold), on any
// ring and stream; errors as text (this also runs on the background thread, which must not touch c.err)
static int pump_produce(Pump* p, hipStream_t st, void* d, size_t bytes, const std::function<void(uint8_t*, size_t, size_t)>& fill, std::string& err,
                        const std::function<void(size_t)>& queued = nullptr /* called with the end of every piece once its copy is queued */) {
    auto bad = [&](hipError_t e, const char* what) { err = std::string(what) + ": " + hipGetErrorString(e); return W2RAP_E_HIP; };
    size_t off = 0; int k = 0;
    while (off < bytes) {
        const int s = k % Pump::NSLOT;
        const size_t n = std::min(Pump::SLOT, bytes - off);
        if (p->used[s]) { const hipError_t e = hipEventSynchronize(p->ev[s]); if (e != hipSuccess) return bad(e, "hipEventSynchronize (staging slot)"); }
        fill(p->slot[s], off, n);
        hipError_t e = hipMemcpyAsync((uint8_t*)d + off, p->slot[s], n, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return bad(e, "hipMemcpyAsync (staging slot -> device)");
        e = hipEventRecord(p->ev[s], st);
        if (e != hipSuccess) return bad(e, "hipEventRecord (staging slot)");
        p->used[s] = true;
        off += n; ++k;
        if (queued) queued(off);
    }
    return 0;
}
// ---- qualities -> one bit per base (q >= min_qual), LSB first: mask byte b holds qualities 8b .. 8b+7
#if defined(__x86_64__)
__attribute__((target("avx2"))) static void mask_piece_avx2(uint8_t* dst, const uint8_t* q, size_t nquals, uint8_t mq) {
    const __m256i t = _mm256_set1_epi8((char)mq);
    size_t i = 0;
    for (; i + 32 <= nquals; i += 32) {
        const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(q + i));
        const uint32_t m = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_max_epu8(v, t), v));      // unsigned v >= t
        std::memcpy(dst + (i >> 3), &m, 4);
    }
    for (; i < nquals; i += 8) {
        unsigned m = 0;
        for (size_t j = 0; j < 8 && i + j < nquals; ++j) m |= (q[i + j] >= mq ? 1u : 0u) << j;
        dst[i >> 3] = (uint8_t)m;
    }
}
#endif
static void mask_piece_plain(uint8_t* dst, const uint8_t* q, size_t nquals, uint8_t mq) {
    for (size_t i = 0; i < nquals; i += 8) {
        unsigned m = 0;
        for (size_t j = 0; j < 8 && i + j < nquals; ++j) m |= (q[i + j] >= mq ? 1u : 0u) << j;
        dst[i >> 3] = (uint8_t)m;
    }
}
// The mask is made by threads of its own into a host buffer the context keeps (huge pages, touched once) WHILE the bases travel -- the
// pool's workers feed the staging ring then --, and goes up behind them like any other array.
struct MaskJob { std::vector<std::thread> th; uint8_t* buf = nullptr; size_t bytes = 0; };
int quality_mask_begin(Ctx& c, const uint8_t* h_quals, uint64_t nq, uint32_t min_qual) {
    const size_t bytes = (size_t)((nq + 7) / 8);
    if (c.h_mask_bytes < bytes) {
        if (c.h_mask) std::free(c.h_mask);
        c.h_mask = host_result_alloc(bytes + bytes / 8);
        c.h_mask_bytes = c.h_mask ? bytes + bytes / 8 : 0;
        if (!c.h_mask) { c.err = "quality mask: out of host memory"; return W2RAP_E_HIP; }
    }
    MaskJob* job = new MaskJob;
    job->buf = static_cast<uint8_t*>(c.h_mask); job->bytes = bytes;
    const uint8_t mq = (uint8_t)std::min<uint32_t>(min_qual, 255u);
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
#endif
    const unsigned hw = std::thread::hardware_concurrency();
    const unsigned nth = hw >= 128 ? 32 : hw >= 64 ? 16 : hw > 8 ? 4 : 1;      // (a single AVX2 stream reads ~6 GB/s of pageable memory: 7.5 GB want many)
    uint8_t* buf = job->buf;
    for (unsigned t = 0; t < nth; ++t)
        job->th.emplace_back([=] {
            const size_t a = (bytes * t / nth) & ~size_t(3), b = t + 1 == nth ? bytes : (bytes * (t + 1) / nth) & ~size_t(3);
            const uint64_t q0 = (uint64_t)a * 8, q1 = std::min<uint64_t>(nq, (uint64_t)b * 8);
            if (q1 <= q0) return;
#if defined(__x86_64__)
            if (avx2) { mask_piece_avx2(buf + a, h_quals + q0, (size_t)(q1 - q0), mq); return; }
#endif
            mask_piece_plain(buf + a, h_quals + q0, (size_t)(q1 - q0), mq);
        });
    c.mask_job = job;
    return 0;
}
void quality_mask_cancel(Ctx& c) {
    if (!c.mask_job) return;
    MaskJob* job = static_cast<MaskJob*>(c.mask_job);
    for (auto& t : job->th) if (t.joinable()) t.join();
    delete job;
    c.mask_job = nullptr;
}
int quality_mask_upload(Ctx& c, uint32_t* d_mask) {
    if (!c.mask_job) { c.err = "quality_mask_upload without quality_mask_begin"; return W2RAP_E_STATE; }
    MaskJob* job = static_cast<MaskJob*>(c.mask_job);
    for (auto& t : job->th) if (t.joinable()) t.join();
    const uint8_t* buf = job->buf; const size_t bytes = job->bytes;
    delete job;
    c.mask_job = nullptr;
    return pump_upload(c, d_mask, buf, bytes);
}
// ---- the raw qualities on the wire: 6 bits each.  A quality is at most 63 (PQVec.cc:30-35: anything above is FATAL in the reference; Q6), so
// four of them travel as three bytes: 7.5 GB of PE150 qualities are 5.6 GB over PCIe -- the late upload is the longest leg of the one-shot
// call.  Layout of a group of four (a, b, c, d in read order): the 24-bit word a << 18 | b << 12 | c << 6 | d, least significant byte first.
// The packed bytes land in a device array of their own and are unpacked by the main stream where it first needs them (k_unpack6; a kernel on
// the copy stream would wait for the counting kernel's CUs and stall the staging ring -- measured).  -> OR of all input bytes (bits 7:6
// set: a value above 63)
static unsigned pack6_plain(uint8_t* dst, const uint8_t* q, size_t n /* multiple of 4 */) {
    unsigned seen = 0;
    for (size_t i = 0; i < n; i += 4) {
        const unsigned a = q[i], b = q[i + 1], c = q[i + 2], d = q[i + 3];
        seen |= a | b | c | d;
        const unsigned v = ((a & 63u) << 18) | ((b & 63u) << 12) | ((c & 63u) << 6) | (d & 63u);
        dst[0] = (uint8_t)v; dst[1] = (uint8_t)(v >> 8); dst[2] = (uint8_t)(v >> 16);
        dst += 3;
    }
    return seen;
}
#if defined(__x86_64__)
__attribute__((target("avx2"))) static unsigned pack6_avx2(uint8_t* dst, const uint8_t* q, size_t n /* multiple of 4 */) {
    // the packing step of a base64 decoder: maddubs merges byte pairs (a * 64 + b), madd merges the 12-bit pairs (x * 4096 + y), a byte shuffle
    // drops the empty byte of every dword, a lane permute closes the gap: 32 qualities -> 24 bytes
    const __m256i m1 = _mm256_set1_epi32(0x01400140), m2 = _mm256_set1_epi32(0x00011000);
    const __m256i shuf = _mm256_setr_epi8(0, 1, 2, 4, 5, 6, 8, 9, 10, 12, 13, 14, -1, -1, -1, -1, 0, 1, 2, 4, 5, 6, 8, 9, 10, 12, 13, 14, -1, -1, -1, -1);
    const __m256i perm = _mm256_setr_epi32(0, 1, 2, 4, 5, 6, 7, 7);
    __m256i acc = _mm256_setzero_si256();
    size_t i = 0;
    for (; i + 64 <= n; i += 32) {                                       // (the 32-byte store writes 8 bytes beyond its 24: only while a later group overwrites them)
        const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(q + i));
        acc = _mm256_or_si256(acc, v);
        const __m256i t = _mm256_madd_epi16(_mm256_maddubs_epi16(_mm256_and_si256(v, _mm256_set1_epi8(63)), m1), m2);
        const __m256i o = _mm256_permutevar8x32_epi32(_mm256_shuffle_epi8(t, shuf), perm);
        _mm256_storeu_si256(reinterpret_cast<__m256i*>(dst + i / 4 * 3), o);
    }
    alignas(32) uint8_t a8[32];
    _mm256_store_si256(reinterpret_cast<__m256i*>(a8), acc);
    unsigned seen = 0;
    for (int k = 0; k < 32; ++k) seen |= a8[k];
    return seen | pack6_plain(dst + i / 4 * 3, q + i, n - i);
}
#endif
static unsigned pack6(uint8_t* dst, const uint8_t* q, size_t n) {
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) return pack6_avx2(dst, q, n);
#endif
    return pack6_plain(dst, q, n);
}
// ngroups groups of four qualities from 3 bytes each.  A thread takes four groups: three dwords in, a uint4 out (the pieces start on
// multiples of 256 qualities: 192 bytes); the last, incomplete quad of a piece byte by byte.
__device__ inline uint32_t unpack6_group(uint32_t v) { return (v >> 18) | (((v >> 12) & 63u) << 8) | (((v >> 6) & 63u) << 16) | ((v & 63u) << 24); }
__global__ void __launch_bounds__(256) k_unpack6(uint64_t ngroups, const uint8_t* __restrict__ src, uint32_t* __restrict__ dst) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, g = 4 * t;
    if (g >= ngroups) return;
    if (g + 4 <= ngroups) {
        const uint32_t* w = reinterpret_cast<const uint32_t*>(src) + 3 * t;
        const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
        uint4 o;
        o.x = unpack6_group(w0 & 0xFFFFFFu); o.y = unpack6_group((w0 >> 24) | ((w1 & 0xFFFFu) << 8));
        o.z = unpack6_group((w1 >> 16) | ((w2 & 0xFFu) << 16)); o.w = unpack6_group(w2 >> 8);
        reinterpret_cast<uint4*>(dst)[t] = o;
    } else {
        for (uint64_t j = g; j < ngroups; ++j)
            dst[j] = unpack6_group((uint32_t)src[3 * j] | ((uint32_t)src[3 * j + 1] << 8) | ((uint32_t)src[3 * j + 2] << 16));
    }
}
constexpr size_t PIECE_Q = Pump::SLOT / 3 * 4 / 256 * 256;                // qualities per piece of the packed upload: their 3/4 fit a staging slot
// ---- the raw qualities behind everything else: a host thread feeds its own staging ring and stream
struct QualsJob {
    std::thread th; hipEvent_t ev = nullptr; int rc = 0; std::string err;
    // a first part of the array that read pathing may start on while the rest still travels
    uint64_t prefix_bytes = 0, prefix_reads = 0; hipEvent_t ev_prefix = nullptr; std::atomic<int> prefix_state{0};      // 0 pending, 1 recorded, -1 none
    // 6-bit-packed upload: the packed bytes land in a device array of their own (copies only: a kernel on the copy stream would wait for the
    // counting kernel's CUs and stall the ring); c.stream unpacks them where it first needs them -- the prefix, then the rest
    bool packed = false; uint8_t* d_packed = nullptr; uint8_t* d_quals = nullptr; uint64_t nq = 0;
    std::atomic<uint64_t> prefix_end{0};             // qualities [0, prefix_end) are up when ev_prefix fires
    uint64_t unpacked = 0;                           // qualities already unpacked on c.stream
};
// unpack qualities [a, b) of the job (a a multiple of 16) on stream st
static void quals_unpack(QualsJob* job, uint64_t a, uint64_t b, hipStream_t st) {
    if (!job->packed || b <= a) return;
    const uint64_t groups = (b - a + 3) / 4;
    hipLaunchKernelGGL(k_unpack6, dim3((unsigned)(((groups + 3) / 4 + 255) / 256)), dim3(256), 0, st, groups, (const uint8_t*)(job->d_packed + a / 4 * 3),
                       reinterpret_cast<uint32_t*>(job->d_quals + a));
}
int quals_upload_begin(Ctx& c, uint8_t* d_quals, const uint8_t* h_quals, uint64_t nq, uint64_t prefix_bytes, uint64_t prefix_reads) {
    if (c.quals_job) { c.err = "quals_upload_begin: an upload is pending"; return W2RAP_E_STATE; }
    if (!c.pump2) {
        Pump* p = new Pump;
        if (p->init()) { p->destroy(); delete p; c.err = "quals_upload_begin: no pinned staging ring"; return W2RAP_E_HIP; }
        c.pump2 = p;
    }
    if (!c.copy_stream) W2_HIP(hipStreamCreateWithFlags(&c.copy_stream, hipStreamNonBlocking));
    // six bits per quality on the wire (above): the pieces land in a small device ring and are unpacked to their place behind their arrival
    // OPT-IN (W2RAP_QUAL_PACK=1).  Measured in round 6 on 50 M PE150 reads, three alternating pairs of runs on one box: the packed bytes take
    // 110 ms on the wire instead of 133, read pathing's wait for the qualities falls from ~50 to ~28 ms -- and the call as a whole does not
    // move (0.244 s packed against 0.242 s, spread 0.22-0.27 either way): 60 ms of AVX2 packing on 16 host threads compete with the staging
    // copies of everything else the call moves through the same host memory system.  Exact either way (tests/test_gpu_boundary.py).
    const char* qp = getenv("W2RAP_QUAL_PACK");
    const bool packed = qp && atoi(qp) != 0 && (reinterpret_cast<uintptr_t>(d_quals) & 15u) == 0;
    if (packed) {
        if (c.d_qring) { c.park(c.d_qring); c.d_qring = nullptr; }
        c.d_qring = c.alloc<uint8_t>((size_t)((nq + 3) / 4 * 3 + 64), false);      // the packed array (until it is unpacked)
        if (!c.d_qring) return W2RAP_E_HIP;
    }
    QualsJob* job = new QualsJob;
    job->packed = packed; job->d_packed = c.d_qring; job->d_quals = d_quals; job->nq = nq;
    if (hipEventCreateWithFlags(&job->ev, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&job->ev_prefix, hipEventDisableTiming) != hipSuccess) {
        delete job; c.err = "quals_upload_begin: hipEventCreate failed"; return W2RAP_E_HIP;
    }
    job->prefix_bytes = prefix_bytes; job->prefix_reads = prefix_reads;
    if (!prefix_bytes || prefix_bytes >= nq) job->prefix_state = -1;
    Pump* p2 = static_cast<Pump*>(c.pump2);
    const int device = c.device; hipStream_t cs = c.copy_stream;
    uint8_t* ring = c.d_qring;
    job->th = std::thread([job, p2, device, cs, d_quals, h_quals, nq, packed, ring] {
        if (hipSetDevice(device) != hipSuccess) { job->rc = W2RAP_E_HIP; job->err = "hipSetDevice failed on the upload thread"; return; }
        const auto t_begin = std::chrono::steady_clock::now();
        double t_pack = 0, t_wait = 0;
        auto prefix = [&](size_t end) {
            if (job->prefix_state.load() == 0 && end >= job->prefix_bytes)
                job->prefix_state = hipEventRecord(job->ev_prefix, cs) == hipSuccess ? 1 : -1;
        };
        if (!packed) {
            job->rc = pump_produce(p2, cs, d_quals, (size_t)nq, [&](uint8_t* dst, size_t off, size_t n) { parallel_memcpy(dst, h_quals + off, n); }, job->err, prefix);
        } else {
            auto bad = [&](hipError_t e, const char* what) { job->err = std::string(what) + ": " + hipGetErrorString(e); job->rc = W2RAP_E_HIP; };
            std::atomic<unsigned> seen{0};
            size_t off = 0; int k = 0;
            while (off < nq && !job->rc) {
                const int s = k % Pump::NSLOT;
                const size_t n = std::min<size_t>(PIECE_Q, nq - off), groups = (n + 3) / 4;
                const auto t0 = std::chrono::steady_clock::now();
                if (p2->used[s]) { const hipError_t e = hipEventSynchronize(p2->ev[s]); if (e != hipSuccess) { bad(e, "hipEventSynchronize (staging slot)"); break; } }
                const auto t1 = std::chrono::steady_clock::now();
                t_wait += std::chrono::duration<double, std::milli>(t1 - t0).count();
                // pack piece [off, off + n) into the pinned slot: the worker threads take stretches of whole groups; the array's last, partial group is padded
                const size_t full = n / 4 * 4, chunk = 1u << 19;
                const size_t nchunks = (full + chunk - 1) / chunk;
                uint8_t* slot = p2->slot[s];
                host_parallel_for_wide(nchunks, [&](size_t i) {
                    const size_t a = i * chunk, b = std::min(full, a + chunk);
                    seen.fetch_or(pack6(slot + a / 4 * 3, h_quals + off + a, b - a));
                });
                if (full < n) { uint8_t t[4] = {0, 0, 0, 0}; for (size_t j = full; j < n; ++j) t[j - full] = h_quals[off + j]; seen.fetch_or(pack6_plain(slot + full / 4 * 3, t, 4)); }
                t_pack += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
                hipError_t e = hipMemcpyAsync(ring + off / 4 * 3, slot, groups * 3, hipMemcpyHostToDevice, cs);
                if (e != hipSuccess) { bad(e, "hipMemcpyAsync (staging slot -> device)"); break; }
                e = hipEventRecord(p2->ev[s], cs);
                if (e != hipSuccess) { bad(e, "hipEventRecord (staging slot)"); break; }
                p2->used[s] = true;
                off += n; ++k;
                if (job->prefix_state.load() == 0 && off >= job->prefix_bytes) job->prefix_end.store(off);
                prefix(off);
            }
            if (!job->rc && (seen.load() & 0xC0u)) {
                job->rc = W2RAP_E_ARG; job->err = "a quality value above 63 (fatal in the reference too: PQVec.cc:30-35)";
            }
        }
        if (!job->rc && hipEventRecord(job->ev, cs) != hipSuccess) { job->rc = W2RAP_E_HIP; job->err = "hipEventRecord failed on the upload thread"; }
        if (getenv("W2RAP_TRACE"))
            fprintf(stderr, "[w2rap] late qualities: %.2f GB %s queued in %.1f ms (packing %.1f ms, waiting for a free staging slot %.1f ms)\n", nq / 1e9, packed ? "packed to 6 bits and" : "",
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(), t_pack, t_wait);
        if (job->prefix_state.load() == 0) job->prefix_state = -1;
    });
    c.quals_job = job;
    return 0;
}
int quals_wait(Ctx& c) {
    if (!c.quals_job) return 0;
    QualsJob* job = static_cast<QualsJob*>(c.quals_job);
    c.quals_job = nullptr;
    if (job->th.joinable()) job->th.join();
    int rc = job->rc;
    if (rc) c.err = "late quality upload: " + job->err;
    else if (hipStreamWaitEvent(c.stream, job->ev, 0) != hipSuccess) { rc = W2RAP_E_HIP; c.err = "late quality upload: hipStreamWaitEvent failed"; }
    if (!rc && job->packed) {
        // everything is up: the part not yet unpacked, on c.stream (13 GB of traffic for 50 M PE150 reads: ~4 ms, against 35 ms of wire saved);
        // the packed array goes back to the pool behind it
        quals_unpack(job, job->unpacked, job->nq, c.stream);
        if (hipStreamSynchronize(c.stream) != hipSuccess) { rc = W2RAP_E_HIP; c.err = "late quality upload: unpacking failed"; }
    }
    if (job->packed && c.d_qring) { if (rc && c.copy_stream) (void)hipStreamSynchronize(c.copy_stream); c.park(c.d_qring); c.d_qring = nullptr; }
    if (rc && c.copy_stream) (void)hipStreamSynchronize(c.copy_stream);
    (void)hipEventDestroy(job->ev);
    (void)hipEventDestroy(job->ev_prefix);
    delete job;
    return rc;
}
// the first part of the qualities: waits (on the host) until its last copy is queued, then lets c.stream wait for it.  -> reads covered (0: none)
int quals_wait_prefix(Ctx& c, uint64_t* reads_covered) {
    *reads_covered = 0;
    if (!c.quals_job) return 0;
    QualsJob* job = static_cast<QualsJob*>(c.quals_job);
    while (job->prefix_state.load() == 0) std::this_thread::sleep_for(std::chrono::microseconds(50));
    if (job->prefix_state.load() != 1) return 0;
    if (hipStreamWaitEvent(c.stream, job->ev_prefix, 0) != hipSuccess) { c.err = "late quality upload: hipStreamWaitEvent failed"; return W2RAP_E_HIP; }
    if (job->packed) {                               // the pieces up to the one that holds the prefix's last quality: unpacked on c.stream, behind their arrival
        const uint64_t upto = job->prefix_end.load() & ~uint64_t(15);
        quals_unpack(job, job->unpacked, upto, c.stream);
        if (upto > job->unpacked) job->unpacked = upto;
        if (job->unpacked < job->prefix_bytes) return 0;          // (cannot happen: the event fires behind the piece that ends at or after the prefix)
    }
    *reads_covered = job->prefix_reads;
    return 0;
}

// device -> host (pageable), complete when it returns
int pump_download(Ctx& c, void* h, const void* d, size_t bytes) {
    if (!bytes) return 0;
    Pump* p = bytes >= (8u << 20) && !getenv("W2RAP_NO_PUMP") ? pump_of(c) : nullptr;
    if (!p) { W2_HIP(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, c.stream)); W2_HIP(hipStreamSynchronize(c.stream)); return 0; }
    for (int s = 0; s < Pump::NSLOT; ++s) if (p->used[s]) { W2_HIP(hipEventSynchronize(p->ev[s])); p->used[s] = false; }
    const size_t npieces = (bytes + Pump::SLOT - 1) / Pump::SLOT;
    for (size_t k = 0; k < npieces + Pump::NSLOT - 1; ++k) {
        if (k < npieces) {                                               // queue piece k
            const int s = (int)(k % Pump::NSLOT);
            const size_t off = k * Pump::SLOT, n = std::min(Pump::SLOT, bytes - off);
            W2_HIP(hipMemcpyAsync(p->slot[s], (const uint8_t*)d + off, n, hipMemcpyDeviceToHost, c.stream));
            W2_HIP(hipEventRecord(p->ev[s], c.stream));
        }
        if (k + 1 >= (size_t)Pump::NSLOT) {                              // drain piece k - (NSLOT-1)
            const size_t j = k + 1 - Pump::NSLOT;
            const int s = (int)(j % Pump::NSLOT);
            const size_t off = j * Pump::SLOT, n = std::min(Pump::SLOT, bytes - off);
            W2_HIP(hipEventSynchronize(p->ev[s]));
            parallel_memcpy((uint8_t*)h + off, p->slot[s], n);
        }
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------ context cache
namespace {
std::mutex g_cache_mu;
std::vector<w2rap_step2_ctx*> g_idle;
}
void drop_reads(Ctx& c);
void drop_results(Ctx& c);

}  // namespace w2

extern "C" {

w2rap_step2_ctx* w2rap_step2_create(int device, char* err, size_t errlen);
void w2rap_step2_destroy(w2rap_step2_ctx*);

// an idle cached context of this device, or a new one
w2rap_step2_ctx* w2rap_step2_acquire(int device, char* err, size_t errlen) {
    {
        std::lock_guard<std::mutex> g(g_cache_mu);
        for (size_t i = 0; i < g_idle.size(); ++i)
            if (g_idle[i]->c.device == device) { w2rap_step2_ctx* h = g_idle[i]; g_idle[i] = g_idle.back(); g_idle.pop_back(); (void)hipSetDevice(device); return h; }
    }
    return w2rap_step2_create(device, err, errlen);
}
// hands a context back: its results and reads are dropped (the device blocks stay parked in its pool), at most two idle contexts per
// device are kept; W2RAP_NO_CTX_CACHE=1 destroys instead
void w2rap_step2_release(w2rap_step2_ctx* h) {
    if (!h) return;
    (void)hipSetDevice(h->c.device);
    drop_results(h->c);
    drop_reads(h->c);
    h->c.err.clear();
    h->c.prof_sums.clear();
    bool keep = !getenv("W2RAP_NO_CTX_CACHE");
    if (keep) {
        std::lock_guard<std::mutex> g(g_cache_mu);
        unsigned same = 0;
        for (auto* x : g_idle) same += x->c.device == h->c.device;
        if (same >= 2) keep = false; else g_idle.push_back(h);
    }
    if (!keep) w2rap_step2_destroy(h);
}
// destroys the idle cached contexts (their pooled device memory goes back to the driver)
int w2rap_step2_trim_cached(void) {
    std::vector<w2rap_step2_ctx*> v;
    { std::lock_guard<std::mutex> g(g_cache_mu); v.swap(g_idle); }
    for (auto* h : v) w2rap_step2_destroy(h);
    return (int)v.size();
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------ the run
namespace {

void set_err(char* err, size_t errlen, const std::string& m) { if (err && errlen) std::snprintf(err, errlen, "%s", m.c_str()); }

int write_freqs(const char* path, const uint64_t* hist, std::string& err) {       // small_K.freqs, BuildReadQGraph.cc:1108-1112
    FILE* f = std::fopen(path, "w");
    if (!f) { err = std::string("cannot write ") + path; return W2RAP_E_IO; }
    for (int i = 1; i < 101; ++i) std::fprintf(f, "%d, %llu\n", i, (unsigned long long)hist[i]);
    std::fclose(f);
    return 0;
}

int run_single(const w2rap_reads* reads, const w2rap_step2_params* p, int device, w2rap_step2_out* out, char* err, size_t errlen) {
    w2rap_step2_ctx* h = w2rap_step2_acquire(device, err, errlen);
    if (!h) return W2RAP_E_NO_DEVICE;
    h->c.n_passes = p->n_passes;
    const bool trace = getenv("W2RAP_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    // host arrays with raw qualities: the quality windows run on a mask made on the way up, the raw bytes follow under the counting
    h->c.hint_min_qual = (reads->mem == W2RAP_MEM_HOST && reads->quals && reads->qual_off) ? (int)std::min<uint32_t>(p->min_qual, 255u) : -1;
    h->c.hint_graph_only = (p->flags & W2RAP_F_GRAPH_ONLY) != 0;
    int rc = w2rap_step2_set_reads(h, reads);
    h->c.hint_min_qual = -1; h->c.hint_graph_only = false;
    const double t1 = now();
    if (!rc) rc = w2rap_step2_count_kmers(h, p->min_qual, p->min_freq, nullptr);
    const double t2 = now();
    if (!rc) rc = w2rap_step2_build_graph(h, p->edge_order_hint);
    const double t3 = now();
    if (!rc && !(p->flags & W2RAP_F_GRAPH_ONLY)) rc = w2rap_step2_path_reads(h);       // pPaths == nullptr: BuildReadQGraph.cc:1300-1307
    const double t4 = now();
    { const int rq = w2::quals_wait(h->c); if (!rc) rc = rq; }          // (the caller's quality array is free again when this call returns, pathed or not)
    if (!rc) rc = w2rap_step2_fetch(h, out);
    if (trace) fprintf(stderr, "[w2rap] w2rap_step2_run: upload %.1f ms, count %.1f, graph %.1f, path %.1f, download %.1f\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3,
                       (t4 - t3) * 1e3, (now() - t4) * 1e3);
    if (!rc && p->freqs_path) rc = write_freqs(p->freqs_path, out->hist, h->c.err);
    if (rc) set_err(err, errlen, h->c.err);
    h->c.n_passes = 0;
    if (rc) { w2rap_step2_free(out); w2rap_step2_destroy(h); }            // a failed context is not cached
    else w2rap_step2_release(h);
    return rc;
}

// A barrier that also AGREES on failure: the flag is read once, under the barrier's mutex, by the last thread to arrive, and every thread
// leaves with that one answer.  (Reading the flag after the barrier instead lets a fast rank fail in the next section before a slow one
// has looked: the slow one returns, the others wait for it at the next barrier forever.)
struct Barrier {
    Barrier(unsigned n, const std::atomic<int>* failed) : n_(n), failed_(failed) {}
    bool wait() {                                     // -> some rank had failed when the last one arrived (the same answer for all)
        std::unique_lock<std::mutex> g(mu_);
        const uint64_t gen = gen_;
        if (++count_ == n_) { count_ = 0; verdict_ = failed_->load() != 0; ++gen_; cv_.notify_all(); }
        else cv_.wait(g, [&] { return gen_ != gen; });
        return verdict_;                              // (the next verdict cannot be written before every thread has left this wait and come back)
    }
    std::mutex mu_; std::condition_variable cv_; unsigned n_, count_ = 0; uint64_t gen_ = 0; bool verdict_ = false; const std::atomic<int>* failed_;
};

struct Rank {
    w2rap_step2_ctx* h = nullptr;
    int dev = 0;
    uint64_t r0 = 0, r1 = 0;
    std::vector<uint64_t> boff, qoff;                 // this shard's offsets, rebased to 0
    uint64_t M = 0;
    std::vector<uint64_t> recs_per_part, kmers_per_part;
    void* d_recs = nullptr; void* d_counts = nullptr; uint64_t nrec = 0;
    uint32_t* d_rcounts = nullptr; uint32_t* d_rrecs = nullptr;
    void *d_hi = nullptr, *d_lo = nullptr, *d_cc = nullptr, *d_cs = nullptr, *d_cn = nullptr;
    uint64_t S = 0, nchunks = 0;
    uint64_t cur_S = 0, cur_C = 0, prev_S = 0, prev_C = 0;     // solid k-mers / chunks this owner has emitted up to the slice just published / appended
    uint64_t cap = 0, ccap = 0, tot = 0, tot_c = 0; bool over = false;      // this rank's dictionary under construction
    hipStream_t copy_stream = nullptr;
    unsigned ns_planned = 0;                          // bucket slices the library plans for this owner's count
    std::vector<void*> staged_blocks;                 // host-staged route: other owners' solid slices, copied into blocks of my own until dict_end
    w2rap_xchg xch{};                                 // the exchange this rank's sharded graph phase has asked for
    void* red_tmp = nullptr; uint64_t red_lo = 0, red_hi = 0;       // its slice of an all-reduce
    w2rap_step2_out stats{};
    w2rap_step2_out out{};
    int rc = 0; std::string err;
};

// ---- the exchanges of the sharded graph phase (include/w2rap_step2.h w2rap_xchg) between the threads of one process: peer copies
struct PeerPtrs { const void* p[64]; };
template <class T>
__global__ void __launch_bounds__(256) k_sum_peers(uint64_t lo, uint64_t hi, unsigned world, PeerPtrs src, T* __restrict__ out) {
    const uint64_t i = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= hi) return;
    T v = 0;
    for (unsigned r = 0; r < world; ++r) v += static_cast<const T*>(src.p[r])[i];
    out[i - lo] = v;
}

// out[i] = in[i] - base (a shard's slice of the job's offset array, rebased to the shard)
__global__ void __launch_bounds__(256) k_rebase(uint64_t n, uint64_t* __restrict__ a, uint64_t base) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] -= base;
}

// ---- copies between the ranks of one process.  Where the driver grants peer access (or both ranks sit on one device) a copy is one
// hipMemcpyAsync on the receiver's stream; where it does not -- hipDeviceCanAccessPeer says no, or enabling fails: a box whose GPUs hang on
// different root complexes without xGMI, IOMMU settings, a container that hides the links -- the same copy is STAGED through pinned host
// memory: the receiver's thread reads the source device into a bounce buffer (two of them, alternating) and queues the second half on its
// own stream.  Slower (PCIe twice), never wrong; the reference's one in-process call has no such failure mode (w2rap-contigger.cc:338), so
// this one must not have it either.  W2RAP_TEST_NO_PEER=1 forces the staged route between ANY two ranks (how a 1-GPU box tests it).
struct Bounce {
    static constexpr size_t SLOT = 32u << 20;
    uint8_t* slot[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool used[2] = {false, false};
    unsigned k = 0;
    int init() {
        for (int i = 0; i < 2; ++i) {
            if (hipHostMalloc((void**)&slot[i], SLOT, hipHostMallocPortable) != hipSuccess) return 1;
            if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) return 1;
        }
        return 0;
    }
    void destroy() {
        for (int i = 0; i < 2; ++i) { if (slot[i]) (void)hipHostFree(slot[i]); if (ev[i]) (void)hipEventDestroy(ev[i]); slot[i] = nullptr; ev[i] = nullptr; used[i] = false; }
    }
};
std::atomic<int> g_last_peer_mode{0};          // of the latest multi-GPU run of this process: 0 none yet, 1 peer copies, 2 host-staged (at least one pair)

// dst (on dst_dev, this thread's current device) <- src (on src_dev); direct: one asynchronous copy on st; otherwise staged through b
int rank_copy(Bounce* b, bool direct, int dst_dev, void* dst, int src_dev, const void* src, size_t bytes, hipStream_t st, std::string& err, Ctx* c = nullptr) {
    if (!bytes) return 0;
    if (direct && c && dst_dev == src_dev && bytes >= (1u << 20)) {
        // both ends on this device (the rank's own share, ranks sharing a GPU): the copy kernel -- hipMemcpyAsync device-to-device runs through SDMA
        if (device_copy_async(*c, dst, src, bytes, st)) { err = c->err; return W2RAP_E_HIP; }
        return 0;
    }
    if (direct) {
        const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) { err = std::string("copy between ranks: ") + hipGetErrorString(e); return W2RAP_E_HIP; }
        return 0;
    }
    if (!b || !b->slot[0]) { err = "copy between ranks: no staging buffers"; return W2RAP_E_HIP; }
    for (size_t off = 0; off < bytes; off += Bounce::SLOT) {
        const unsigned s = b->k++ & 1u;
        const size_t n = std::min(Bounce::SLOT, bytes - off);
        hipError_t e = hipSuccess;
        if (b->used[s]) e = hipEventSynchronize(b->ev[s]);                 // the previous use of this slot has left for the device
        if (e == hipSuccess) e = hipSetDevice(src_dev);
        if (e == hipSuccess) e = hipMemcpy(b->slot[s], (const uint8_t*)src + off, n, hipMemcpyDeviceToHost);
        const hipError_t e2 = hipSetDevice(dst_dev);
        if (e == hipSuccess) e = e2;
        if (e == hipSuccess) e = hipMemcpyAsync((uint8_t*)dst + off, b->slot[s], n, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipEventRecord(b->ev[s], st);
        if (e != hipSuccess) { err = std::string("host-staged copy between ranks: ") + hipGetErrorString(e); return W2RAP_E_HIP; }
        b->used[s] = true;
    }
    return 0;
}

int run_multi(const w2rap_reads* reads, const w2rap_step2_params* p, unsigned world, const int* devs, w2rap_step2_out* out, char* err, size_t errlen) {
    if (reads->mem != W2RAP_MEM_HOST && reads->mem != W2RAP_MEM_DEVICE) { set_err(err, errlen, "w2rap_step2_run: reads.mem must be W2RAP_MEM_HOST or W2RAP_MEM_DEVICE"); return W2RAP_E_ARG; }
    const bool dev_reads = reads->mem == W2RAP_MEM_DEVICE;       // the arrays live on ONE device (where Step 1 left them, say): every rank takes its shard from there
    const uint64_t n = reads->n_reads;
    const bool raw = reads->quals && reads->qual_off;
    if (n && (!reads->bases_packed || !reads->base_byte_off || !reads->read_len || (raw == (reads->pq && reads->pq_off)))) {
        set_err(err, errlen, "w2rap_step2_run: null base arrays, or not exactly one of (quals, qual_off) and (pq, pq_off)"); return W2RAP_E_ARG;
    }
    int src_dev = -1;                                             // device-resident reads: the device that holds them
    if (dev_reads && n) {
        hipPointerAttribute_t at{};
        if (hipPointerGetAttributes(&at, reads->bases_packed) != hipSuccess) { (void)hipGetLastError(); set_err(err, errlen, "w2rap_step2_run: reads.mem is W2RAP_MEM_DEVICE but bases_packed is not a device pointer"); return W2RAP_E_ARG; }
        src_dev = at.device;
    }
    std::vector<Rank> R(world);
    // contexts first (on the calling thread: errors are plain), peer access between every pair of distinct devices
    for (unsigned r = 0; r < world; ++r) {
        R[r].dev = devs[r];
        R[r].h = w2rap_step2_acquire(devs[r], err, errlen);
        if (!R[r].h) { for (unsigned q = 0; q < r; ++q) w2rap_step2_release(R[q].h); return W2RAP_E_NO_DEVICE; }
    }
    // which pairs of devices reach each other directly; the others are served by host-staged copies (rank_copy)
    const bool force_staged = test_hook("W2RAP_TEST_NO_PEER");
    std::vector<int> alldev(devs, devs + world);
    if (src_dev >= 0) alldev.push_back(src_dev);
    int maxdev = 0; for (int d : alldev) maxdev = std::max(maxdev, d);
    std::vector<std::vector<char>> reach(maxdev + 1, std::vector<char>(maxdev + 1, 0));
    bool any_staged = force_staged;
    std::string why_staged;
    for (int a : alldev) for (int b : alldev) {
        if (a == b) { reach[a][b] = 1; continue; }
        if (reach[a][b]) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, a, b) != hipSuccess) { (void)hipGetLastError(); can = 0; }
        if (can) {
            (void)hipSetDevice(a);
            const hipError_t e = hipDeviceEnablePeerAccess(b, 0);
            (void)hipGetLastError();
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { can = 0; if (why_staged.empty()) why_staged = std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e); }
        } else if (why_staged.empty()) why_staged = "hipDeviceCanAccessPeer(" + std::to_string(a) + ", " + std::to_string(b) + ") says no";
        reach[a][b] = (char)can;
        if (!can) any_staged = true;
    }
    // (a copy is direct when BOTH directions are granted: the receiver reads the sender's memory and kernels read peers in place)
    auto direct_dev = [&](int a, int b) { return a == b || (reach[a][b] && reach[b][a]); };
    auto direct = [&](unsigned a, unsigned b) { return a == b || (!force_staged && direct_dev(devs[a], devs[b])); };
    if (any_staged)
        fprintf(stderr, "[w2rap] w2rap_step2_run: no peer access between some of the %u GPUs (%s): their exchanges are staged through pinned host memory -- "
                        "correct, but at PCIe rates (peer_access: host-staged)\n", world, force_staged ? "W2RAP_TEST_NO_PEER" : why_staged.c_str());
    g_last_peer_mode.store(any_staged ? 2 : 1);
    std::vector<Bounce> bounce(world);
    if (any_staged)
        for (unsigned r = 0; r < world; ++r) {
            (void)hipSetDevice(devs[r]);
            if (bounce[r].init()) {
                for (auto& b : bounce) b.destroy();
                for (auto& x : R) w2rap_step2_release(x.h);
                set_err(err, errlen, "w2rap_step2_run: no pinned host memory for the staged exchanges"); return W2RAP_E_HIP;
            }
        }
    // device-resident reads: the shard boundaries' offsets (two words per rank) come down once
    std::vector<uint64_t> cut_b(world + 1, 0), cut_q(world + 1, 0);
    // shards: contiguous ranges of whole pairs (reads 2i, 2i+1 are mates, ExtractReads.cc:474)
    const uint64_t pairs = n / 2;
    for (unsigned r = 0; r < world; ++r) {
        R[r].r0 = 2 * (pairs * r / world);
        R[r].r1 = r + 1 == world ? n : 2 * (pairs * (r + 1) / world);
    }
    if (dev_reads && n) {
        (void)hipSetDevice(src_dev);
        (void)hipDeviceSynchronize();                                     // the caller's own streams may still be filling the arrays
        const uint64_t* qo = raw ? reads->qual_off : reads->pq_off;
        hipError_t e = hipSuccess;
        for (unsigned r = 0; r <= world && e == hipSuccess; ++r) {
            const uint64_t at = r < world ? R[r].r0 : n;
            e = hipMemcpy(&cut_b[r], reads->base_byte_off + at, 8, hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(&cut_q[r], qo + at, 8, hipMemcpyDeviceToHost);
        }
        if (e != hipSuccess) {
            for (auto& b : bounce) b.destroy();
            for (auto& x : R) w2rap_step2_release(x.h);
            set_err(err, errlen, std::string("w2rap_step2_run: reading the offsets of device-resident reads: ") + hipGetErrorString(e)); return W2RAP_E_HIP;
        }
    }
    std::atomic<int> failed{0};
    Barrier bar(world, &failed);
    uint64_t M_total = 0, D_total = 0, S_total = 0, C_total = 0;
    uint64_t hist[101] = {0};
    uint32_t nb = 0, nbl = 0;
    const unsigned P = p->n_passes > 1 ? p->n_passes : 1;       // hash-range passes of the counting phase (0 and 1: one pass; the owners already divide the records by n_gpus)
    const char* rg = getenv("W2RAP_REPLICATED_GRAPH");                          // "1" (any non-zero number): the gathered dictionary of rounds 1-4; "0" / empty: sharded, like unset
    const bool sharded = !(p->flags & W2RAP_F_REPLICATED_GRAPH) && !(rg && atoi(rg) != 0);   // row e-3: dictionary, prune, unipaths stay with their owners

    int fail_rank = -1, fail_stage = 0;
    if (const char* fv = getenv("W2RAP_TEST_FAIL_AT")) { if (test_hook("W2RAP_TEST_FAIL_AT")) std::sscanf(fv, "%d:%d", &fail_rank, &fail_stage); }
    auto body = [&](unsigned me) {
        Rank& X = R[me];
        w2rap_step2_ctx* h = X.h;
        Ctx& c = h->c;
        auto fail = [&](int rc, const std::string& m) { X.rc = rc; X.err = m.empty() ? c.err : m; failed.store(1); };
        auto check = [&](int rc) { if (rc && !X.rc) fail(rc, ""); };
        // test hook W2RAP_TEST_FAIL_AT="rank:stage": that rank fails at that stage (1 partition, 2 shuffle, 3 counting, 4 dictionary) -- every
        // other rank must come back with an error too instead of waiting at a barrier for ever
        auto inject = [&](int stage) {
            if (fail_rank == (int)me && fail_stage == stage && !X.rc) fail(W2RAP_E_HIP, "injected failure (W2RAP_TEST_FAIL_AT)");
        };
        (void)hipSetDevice(X.dev);
        if (hipStreamCreateWithFlags(&X.copy_stream, hipStreamNonBlocking) != hipSuccess) fail(W2RAP_E_HIP, "hipStreamCreate failed");
        Bounce* bb = any_staged ? &bounce[me] : nullptr;
        std::string cerr_;
        // a copy from rank r's memory into mine, on stream st
        auto pull = [&](void* dst, unsigned r, const void* src, size_t bytes, hipStream_t st) {
            if (X.rc) return;
            if (rank_copy(bb, direct(me, r), X.dev, dst, R[r].dev, src, bytes, st, cerr_, &c)) fail(W2RAP_E_HIP, cerr_);
        };
        // ---- A: this shard's reads, quality windows
        if (dev_reads) {
            // the shard's pieces of the job's device arrays -> arrays of this rank's own (its context's pool), offsets rebased to the shard
            const uint64_t m = X.r1 - X.r0;
            const uint64_t b0 = cut_b[me], b1 = cut_b[me + 1], q0 = cut_q[me], q1 = cut_q[me + 1];
            const bool dsrc = !force_staged ? direct_dev(X.dev, src_dev) : X.dev == src_dev;      // (the hook: staged unless it is this rank's own device)
            // (untracked blocks: set_reads drops the context's results -- everything tracked -- before it takes the arrays; they join the
            //  context's reads behind it and go back to the pool with them)
            uint8_t* l_bases = c.alloc<uint8_t>(b1 - b0 + 64, false); uint64_t* l_boff = c.alloc<uint64_t>(m + 2, false); uint32_t* l_len = c.alloc<uint32_t>(m + 1, false);
            uint8_t* l_q = c.alloc<uint8_t>(q1 - q0 + 64, false); uint64_t* l_qoff = c.alloc<uint64_t>(m + 2, false);
            if (!l_bases || !l_boff || !l_len || !l_q || !l_qoff) fail(W2RAP_E_HIP, "");
            auto take = [&](void* dst, const void* src, size_t bytes) {
                if (!X.rc && rank_copy(bb, dsrc, X.dev, dst, src_dev, src, bytes, c.stream, cerr_)) fail(W2RAP_E_HIP, cerr_);
            };
            if (!X.rc && m) {
                const uint64_t* qo = raw ? reads->qual_off : reads->pq_off;
                take(l_bases, reads->bases_packed + b0, b1 - b0);
                take(l_boff, reads->base_byte_off + X.r0, (m + 1) * 8);
                take(l_len, reads->read_len + X.r0, m * 4);
                take(l_q, (raw ? reads->quals : reads->pq) + q0, q1 - q0);
                take(l_qoff, qo + X.r0, (m + 1) * 8);
                if (!X.rc) {
                    hipLaunchKernelGGL(k_rebase, dim3((unsigned)((m + 256) / 256)), dim3(256), 0, c.stream, m + 1, l_boff, b0);
                    hipLaunchKernelGGL(k_rebase, dim3((unsigned)((m + 256) / 256)), dim3(256), 0, c.stream, m + 1, l_qoff, q0);
                    if (hipMemsetAsync(l_bases + (b1 - b0), 0, 64, c.stream) != hipSuccess || hipMemsetAsync(l_q + (q1 - q0), 0, 64, c.stream) != hipSuccess ||
                        hipStreamSynchronize(c.stream) != hipSuccess) fail(W2RAP_E_HIP, "taking the shard of device-resident reads failed");
                }
            }
            w2rap_reads s{};
            s.n_reads = m; s.mem = W2RAP_MEM_DEVICE;
            s.bases_packed = l_bases; s.base_byte_off = l_boff; s.read_len = l_len;
            if (raw) { s.quals = l_q; s.qual_off = l_qoff; } else { s.pq = l_q; s.pq_off = l_qoff; }
            if (!X.rc) check(w2rap_step2_set_reads(h, &s));
            for (void* q : {(void*)l_bases, (void*)l_boff, (void*)l_len, (void*)l_q, (void*)l_qoff}) if (q) c.owned_reads.push_back(q);
            if (!X.rc) check(w2rap_step2_quality_windows(h, p->min_qual, &X.M));
        } else {
            const uint64_t m = X.r1 - X.r0;
            w2rap_reads s{};
            s.n_reads = m; s.mem = W2RAP_MEM_HOST;
            X.boff.assign(m + 1, 0); X.qoff.assign(m + 1, 0);
            const uint64_t b0 = m ? reads->base_byte_off[X.r0] : 0;
            for (uint64_t i = 0; i <= m && m; ++i) X.boff[i] = reads->base_byte_off[X.r0 + i] - b0;
            s.bases_packed = m ? reads->bases_packed + b0 : reads->bases_packed; s.base_byte_off = X.boff.data(); s.read_len = m ? reads->read_len + X.r0 : reads->read_len;
            const uint64_t* qo = raw ? reads->qual_off : reads->pq_off;
            const uint64_t q0 = m ? qo[X.r0] : 0;
            for (uint64_t i = 0; i <= m && m; ++i) X.qoff[i] = qo[X.r0 + i] - q0;
            if (raw) { s.quals = reads->quals + q0; s.qual_off = X.qoff.data(); } else { s.pq = reads->pq + q0; s.pq_off = X.qoff.data(); }
            check(w2rap_step2_set_reads(h, &s));
            if (!X.rc) check(w2rap_step2_quality_windows(h, p->min_qual, &X.M));
        }
        if (bar.wait()) return;
        if (me == 0) {
            M_total = 0; for (auto& y : R) M_total += y.M;
            nb = w2rap_step2_default_buckets(M_total, world * P); nbl = nb / world / P;
        }
        bar.wait();
        // the owner's k-mer instances over ALL passes bound its solid set (S <= instances / min_freq): buckets are hash-uniform, so a
        // generous share of the job's instances; a small job simply takes all of them
        // (with several passes the solid arrays are sized in pass 0 for ALL passes: from pass 0's own share -- buckets are hash-uniform, so every
        // pass brings this owner about as many instances -- with a quarter of head room, not from a blanket "twice the mean": 20 B per entry)
        for (unsigned pass = 0; pass < P; ++pass) {
            // ---- B: super-k-mer records of the shard for this pass's bucket range (MapReduceEngine.h:288-299: the reads are cut again in every
            //      pass, records of other ranges are dropped), grouped by bucket = by owner
            const uint32_t lo = nb / P * pass, hi = lo + nb / P;
            X.recs_per_part.assign(world, 0); X.kmers_per_part.assign(world, 0);
            check(w2rap_step2_partition_range(h, nb, lo, hi, world, X.recs_per_part.data(), X.kmers_per_part.data()));
            if (!X.rc) check(w2rap_step2_partition_buffers(h, &X.d_recs, &X.d_counts, &X.nrec));
            inject(1);
            if (bar.wait()) return;
            // ---- C: the k-mer shuffle, bucket slice by bucket slice (dist.py's pipeline inside the one in-process call).  Owner `me` pulls the
            //      per-bucket counts of its range from every source, plans the count in slices, and queues the record rows of slice 0, 1, ..
            //      on a copy stream: slice k is counted as soon as ITS rows have arrived, slice k+1 travels meanwhile.
            uint64_t owned_kmers = 0, rows = 0;
            for (auto& y : R) { owned_kmers += y.kmers_per_part[me]; rows += y.recs_per_part[me]; }
            X.d_rcounts = c.alloc<uint32_t>((uint64_t)world * nbl);
            X.d_rrecs = c.alloc<uint32_t>(rows * REC_DWORDS + 1);
            if (!X.d_rcounts || !X.d_rrecs) fail(W2RAP_E_HIP, "");
            std::vector<uint32_t> hc((size_t)world * nbl);
            if (!X.rc) {
                for (unsigned s = 0; s < world && !X.rc; ++s)
                    pull(X.d_rcounts + (uint64_t)s * nbl, s, (const uint32_t*)R[s].d_counts + (uint64_t)me * nbl, (size_t)nbl * 4, c.stream);
                if (!X.rc && (hipMemcpyAsync(hc.data(), X.d_rcounts, hc.size() * 4, hipMemcpyDeviceToHost, c.stream) != hipSuccess || hipStreamSynchronize(c.stream) != hipSuccess))
                    fail(W2RAP_E_HIP, "peer copy of the bucket counts failed");
            }
            inject(2);
            if (!X.rc && P > 1) check(w2rap_step2_count_pass(h, pass, P));
            const uint64_t owned_bound = M_total < (1ull << 24) ? M_total : std::min<uint64_t>(M_total, owned_kmers * P + owned_kmers * P / 4 + (1ull << 20));
            if (!X.rc) check(w2rap_step2_count_records_begin(h, p->min_freq, nbl, world, X.d_rrecs, X.d_rcounts, P > 1 ? owned_bound : owned_kmers, 4, 1));
            // the library's own slice plan for nbl buckets (the same on every rank: it depends on nbl alone)
            const unsigned ns = X.rc ? 1u : (unsigned)w2rap_step2_count_records_slices(h);
            X.ns_planned = ns;
            if (!X.rc && (ns < 1 || ns > 16)) fail(W2RAP_E_STATE, "unexpected number of bucket slices");
            if (bar.wait()) return;
            for (auto& y : R) if (y.ns_planned != ns && !X.rc) fail(W2RAP_E_STATE, "the ranks plan different numbers of bucket slices");
            if (!X.rc && pass == 0) {                                     // the arrays the solid k-mers of every slice and pass are appended to
                check(w2rap_step2_solid_buffers(h, &X.d_hi, &X.d_lo, &X.d_cc, nullptr));
                if (!X.rc) check(w2rap_step2_chunk_buffers(h, &X.d_cs, &X.d_cn, nullptr));
            }
            hipEvent_t ev[16] = {};
            if (!X.rc) {
                std::vector<uint32_t> cut(ns + 1, 0);
                for (unsigned k = 0; k < ns && !X.rc; ++k) { uint32_t lo_b = 0, hi_b = 0; check(w2rap_step2_count_records_bounds(h, k, &lo_b, &hi_b)); cut[k] = lo_b; cut[k + 1] = hi_b; }
                // rows of source s before bucket b of my range; the segments of the sources lie back to back at the owner
                std::vector<std::vector<uint64_t>> pre(world, std::vector<uint64_t>(ns + 1, 0));
                for (unsigned s = 0; s < world; ++s) {
                    uint64_t run = 0; unsigned k = 0;
                    for (uint32_t bkt = 0; bkt <= nbl; ++bkt) {
                        while (k <= ns && cut[k] == bkt) pre[s][k++] = run;
                        if (bkt < nbl) run += hc[(size_t)s * nbl + bkt];
                    }
                }
                for (unsigned k = 0; k < ns && !X.rc; ++k) {
                    uint64_t seg = 0;
                    for (unsigned s = 0; s < world && !X.rc; ++s) {
                        Rank& Y = R[s];
                        uint64_t before = 0; for (unsigned q = 0; q < me; ++q) before += Y.recs_per_part[q];
                        const uint64_t a = pre[s][k], e = pre[s][k + 1];
                        if (e > a) pull(X.d_rrecs + (seg + a) * REC_DWORDS, s, (const uint32_t*)Y.d_recs + (before + a) * REC_DWORDS, (e - a) * REC_BYTES, X.copy_stream);
                        seg += Y.recs_per_part[me];
                    }
                    if (!X.rc && (hipEventCreateWithFlags(&ev[k], hipEventDisableTiming) != hipSuccess || hipEventRecord(ev[k], X.copy_stream) != hipSuccess))
                        fail(W2RAP_E_HIP, "event on the copy stream failed");
                    // ---- D: slice k counts behind the arrival of its rows (queued at once: with peer copies every slice's rows are in flight
                    //      before the first count starts; with staged copies slice k counts while this thread carries slice k+1)
                    if (!X.rc && hipStreamWaitEvent(c.stream, ev[k], 0) != hipSuccess) fail(W2RAP_E_HIP, "hipStreamWaitEvent failed");
                    if (!X.rc) check(w2rap_step2_count_records_launch(h, k));
                }
            }
            // ---- E: while slice k+1 counts, every rank appends slice k's solid k-mers of EVERY owner to its dictionary (peer copies and
            //      inserts on the library's side stream), owners in rank order: identical k-mer numbering on every rank
            for (unsigned k = 0; k < ns; ++k) {
                uint64_t sk = 0, ck = 0;
                if (k == 1 || ns == 1) inject(3);
                if (!X.rc) check(w2rap_step2_count_records_slice(h, k, &sk, &ck));
                X.cur_S = sk; X.cur_C = ck;
                if (bar.wait()) { for (auto& e : ev) if (e) (void)hipEventDestroy(e); return; }      // every owner's slice k is counted and published
                uint64_t n_all = 0, c_all = 0;
                for (auto& y : R) { n_all += y.cur_S - y.prev_S; c_all += y.cur_C - y.prev_C; }
                if (sharded && k + 1 == ns) inject(4);
                if (sharded && !X.rc) {
                    // the owner's own dictionary takes slice k's solid k-mers on the side stream while slice k+1 counts
                    if (pass == 0 && k == 0) {
                        uint32_t lo_b = 0, hi_b = 1;
                        check(w2rap_step2_count_records_bounds(h, 0, &lo_b, &hi_b));
                        X.cap = (uint64_t)((double)sk * ((double)nbl * P / std::max<uint32_t>(hi_b - lo_b, 1)) * 1.15) + 4096;
                    }
                    if (!X.rc) check(w2rap_step2_local_dict_slice(h, sk, X.cap));
                }
                if (sharded) { if (bar.wait()) { for (auto& e : ev) if (e) (void)hipEventDestroy(e); return; } X.prev_S = X.cur_S; X.prev_C = X.cur_C; continue; }
                if (pass == 0 && k == 0) {
                    // buckets are hash-uniform: the first slice predicts the whole (with head room); a wrong guess falls back to the whole-set gather
                    uint32_t lo_b = 0, hi_b = 1;
                    if (!X.rc) check(w2rap_step2_count_records_bounds(h, 0, &lo_b, &hi_b));
                    const double scale = (double)nbl * P / std::max<uint32_t>(hi_b - lo_b, 1) * 1.15;
                    X.cap = std::min<uint64_t>((uint64_t)(n_all * scale) + 4096, MAX_SOLID_KMERS - 1); X.ccap = (uint64_t)(c_all * scale) + 4096;
                    if (test_hook("W2RAP_TEST_SMALL_DICT")) { X.cap = n_all + 1; }
                    if (!X.rc) check(w2rap_step2_dict_begin(h, X.cap, X.ccap));
                }
                if (k + 1 == ns) inject(4);
                if (X.tot + n_all > X.cap || X.tot_c + c_all > X.ccap) X.over = true;       // (the same decision on every rank: all see the same totals)
                if (!X.over) {
                    for (unsigned o = 0; o < world && !X.rc; ++o) {
                        Rank& Y = R[o];
                        const uint64_t n = Y.cur_S - Y.prev_S, nc = Y.cur_C - Y.prev_C;
                        const uint64_t* a_hi = (const uint64_t*)Y.d_hi + Y.prev_S; const uint64_t* a_lo = (const uint64_t*)Y.d_lo + Y.prev_S; const uint32_t* a_cc = (const uint32_t*)Y.d_cc + Y.prev_S;
                        const uint64_t* a_cs = nc ? (const uint64_t*)Y.d_cs + Y.prev_C : nullptr; const uint32_t* a_cn = nc ? (const uint32_t*)Y.d_cn + Y.prev_C : nullptr;
                        if (!direct(me, o) && n) {
                            // no peer access to that owner: its slice comes over into blocks of my own, which the dictionary reads until dict_end
                            uint64_t* t_hi = c.alloc<uint64_t>(n); uint64_t* t_lo = c.alloc<uint64_t>(n); uint32_t* t_cc = c.alloc<uint32_t>(n);
                            uint64_t* t_cs = c.alloc<uint64_t>(nc + 1); uint32_t* t_cn = c.alloc<uint32_t>(nc + 1);
                            if (!t_hi || !t_lo || !t_cc || !t_cs || !t_cn) { fail(W2RAP_E_HIP, ""); break; }
                            for (void* q : {(void*)t_hi, (void*)t_lo, (void*)t_cc, (void*)t_cs, (void*)t_cn}) X.staged_blocks.push_back(q);
                            pull(t_hi, o, a_hi, n * 8, c.stream); pull(t_lo, o, a_lo, n * 8, c.stream); pull(t_cc, o, a_cc, n * 4, c.stream);
                            if (nc) { pull(t_cs, o, a_cs, nc * 8, c.stream); pull(t_cn, o, a_cn, nc * 4, c.stream); }
                            if (!X.rc && hipStreamSynchronize(c.stream) != hipSuccess) fail(W2RAP_E_HIP, "staged gather of the solid k-mers failed");
                            a_hi = t_hi; a_lo = t_lo; a_cc = t_cc; a_cs = nc ? t_cs : nullptr; a_cn = nc ? t_cn : nullptr;
                        }
                        if (!X.rc) check(w2rap_step2_dict_append_slice(h, a_hi, a_lo, a_cc, n, a_cs, a_cn, nc, Y.prev_S));
                    }
                    X.tot += n_all; X.tot_c += c_all;
                }
                if (bar.wait()) { for (auto& e : ev) if (e) (void)hipEventDestroy(e); return; }      // everyone has read the published counts
                X.prev_S = X.cur_S; X.prev_C = X.cur_C;
            }
            if (!X.rc && hipStreamSynchronize(X.copy_stream) != hipSuccess) fail(W2RAP_E_HIP, "peer copy of super-k-mer records failed");
            for (auto& e : ev) if (e) (void)hipEventDestroy(e);
            if (!X.rc) check(w2rap_step2_count_records_end(h, &X.stats));
            if (bar.wait()) return;                                       // every owner has its records: the sources' buffers are free
            if (X.d_rcounts) c.release(X.d_rcounts);
            if (X.d_rrecs) c.release(X.d_rrecs);
            X.d_rcounts = nullptr; X.d_rrecs = nullptr;
        }
        X.S = X.cur_S; X.nchunks = X.cur_C;
        if (bar.wait()) return;
        if (me == 0) {
            D_total = S_total = C_total = 0; std::memset(hist, 0, sizeof hist);
            for (auto& y : R) { D_total += y.stats.n_kmers_distinct; S_total += y.S; C_total += y.nchunks; for (int i = 0; i < 101; ++i) hist[i] += y.stats.hist[i]; }
        }
        bar.wait();
        if (sharded) {
            // ---- F': every owner keeps its k-mers; the graph phase runs as a state machine between exchanges, which are peer copies here
            std::vector<uint64_t> spr(world);
            for (unsigned r = 0; r < world; ++r) spr[r] = R[r].S;
            if (!X.rc) check(w2rap_step2_shard_begin(h, me, world, spr.data(), nb, P, M_total, D_total, hist, p->edge_order_hint));
            for (;;) {
                if (!X.rc) check(w2rap_step2_shard_next(h, &X.xch));
                if (bar.wait()) return;                                   // every rank has published its exchange (or one has failed)
                const int op = X.xch.op;
                const uint32_t eb = X.xch.elem_bytes;
                for (auto& y : R) if (y.xch.op != op) fail(W2RAP_E_STATE, "sharded graph: the ranks disagree about the next exchange");
                if (op == W2RAP_X_DONE) { bar.wait(); break; }
                if (op == W2RAP_X_ALLTOALL || op == W2RAP_X_ALLGATHER) {
                    std::vector<uint64_t> rcnt(world), soff(world, 0);
                    for (unsigned r = 0; r < world; ++r) {
                        rcnt[r] = op == W2RAP_X_ALLTOALL ? R[r].xch.send_count[me] : R[r].xch.send_count[0];
                        if (op == W2RAP_X_ALLTOALL) for (unsigned q = 0; q < me; ++q) soff[r] += R[r].xch.send_count[q];
                    }
                    void* buf = nullptr;
                    if (!X.rc) check(w2rap_step2_shard_recv(h, rcnt.data(), eb, &buf));
                    uint64_t at = 0;
                    for (unsigned r = 0; r < world && !X.rc; ++r) {
                        if (rcnt[r]) pull((uint8_t*)buf + at * eb, r, (const uint8_t*)R[r].xch.send + soff[r] * eb, rcnt[r] * eb, c.stream);
                        at += rcnt[r];
                    }
                    if (!X.rc && hipStreamSynchronize(c.stream) != hipSuccess) fail(W2RAP_E_HIP, "peer copy of a sharded-graph exchange failed");
                } else if (op == W2RAP_X_ALLGATHER_HOST) {
                    std::vector<uint64_t> words(world);
                    for (unsigned r = 0; r < world; ++r) words[r] = *static_cast<const uint64_t*>(R[r].xch.send);
                    if (!X.rc) check(w2rap_step2_shard_host_words(h, words.data()));
                } else if (op == W2RAP_X_ALLREDUCE_U8 || op == W2RAP_X_ALLREDUCE_U32) {
                    // reduce-scatter + all-gather by hand: rank `me` sums its slice of every rank's array (a kernel reading the peers), then
                    // everybody collects the finished slices
                    const uint64_t n = X.xch.send_count[0];
                    X.red_lo = n * me / world; X.red_hi = n * (me + 1) / world;
                    const uint64_t m = X.red_hi - X.red_lo;
                    X.red_tmp = c.alloc<uint8_t>(m * eb + 16);
                    if (!X.red_tmp) fail(W2RAP_E_HIP, "");
                    PeerPtrs pp{};
                    std::vector<void*> red_in;
                    for (unsigned r = 0; r < world; ++r) {
                        pp.p[r] = R[r].xch.send;
                        if (direct(me, r) || !m || X.rc) continue;
                        // no peer access: rank r's part of my slice comes over first; the kernel indexes from the array's start
                        uint8_t* t = c.alloc<uint8_t>(m * eb + 16);
                        if (!t) { fail(W2RAP_E_HIP, ""); break; }
                        red_in.push_back(t);
                        pull(t, r, (const uint8_t*)R[r].xch.send + X.red_lo * eb, m * eb, c.stream);
                        pp.p[r] = t - X.red_lo * eb;
                    }
                    if (!X.rc && m) {
                        if (op == W2RAP_X_ALLREDUCE_U8) hipLaunchKernelGGL(k_sum_peers<uint8_t>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c.stream, X.red_lo, X.red_hi, world, pp, (uint8_t*)X.red_tmp);
                        else hipLaunchKernelGGL(k_sum_peers<uint32_t>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c.stream, X.red_lo, X.red_hi, world, pp, (uint32_t*)X.red_tmp);
                    }
                    if (!X.rc && hipStreamSynchronize(c.stream) != hipSuccess) fail(W2RAP_E_HIP, "all-reduce of a sharded-graph exchange failed");
                    for (void* t : red_in) c.release(t);
                    if (bar.wait()) return;                               // every slice is summed: nobody reads the inputs any more
                    for (unsigned r = 0; r < world && !X.rc; ++r) {
                        const uint64_t mr = R[r].red_hi - R[r].red_lo;
                        if (mr) pull((uint8_t*)X.xch.send + R[r].red_lo * eb, r, R[r].red_tmp, mr * eb, c.stream);
                    }
                    if (!X.rc && hipStreamSynchronize(c.stream) != hipSuccess) fail(W2RAP_E_HIP, "all-reduce of a sharded-graph exchange failed");
                    if (bar.wait()) return;                               // the slices have been collected
                    if (X.red_tmp) { c.release(X.red_tmp); X.red_tmp = nullptr; }
                } else fail(W2RAP_E_STATE, "sharded graph: unknown exchange");
                if (bar.wait()) return;                                   // the exchange is complete on every rank: the send buffers are free
            }
            if (!X.rc && !(p->flags & W2RAP_F_GRAPH_ONLY)) check(w2rap_step2_path_reads(h));
            if (!X.rc) check(w2rap_step2_fetch(h, &X.out));
            return;
        }
        if (X.over) {
            // the capacity guessed from the first slice was too small: the whole-set gather, owners in rank order
            if (!X.rc) (void)w2rap_step2_dict_abort(h);
            if (!X.rc) check(w2rap_step2_dict_begin(h, S_total + 1, C_total + 1));
            for (unsigned o = 0; o < world && !X.rc; ++o) {
                Rank& Y = R[o];
                const void *a_hi = Y.d_hi, *a_lo = Y.d_lo, *a_cc = Y.d_cc, *a_cs = Y.d_cs, *a_cn = Y.d_cn;
                if (!direct(me, o) && Y.S) {
                    uint64_t* t_hi = c.alloc<uint64_t>(Y.S); uint64_t* t_lo = c.alloc<uint64_t>(Y.S); uint32_t* t_cc = c.alloc<uint32_t>(Y.S);
                    uint64_t* t_cs = c.alloc<uint64_t>(Y.nchunks + 1); uint32_t* t_cn = c.alloc<uint32_t>(Y.nchunks + 1);
                    if (!t_hi || !t_lo || !t_cc || !t_cs || !t_cn) { fail(W2RAP_E_HIP, ""); break; }
                    for (void* q : {(void*)t_hi, (void*)t_lo, (void*)t_cc, (void*)t_cs, (void*)t_cn}) X.staged_blocks.push_back(q);
                    pull(t_hi, o, a_hi, Y.S * 8, c.stream); pull(t_lo, o, a_lo, Y.S * 8, c.stream); pull(t_cc, o, a_cc, Y.S * 4, c.stream);
                    if (Y.nchunks) { pull(t_cs, o, a_cs, Y.nchunks * 8, c.stream); pull(t_cn, o, a_cn, Y.nchunks * 4, c.stream); }
                    if (!X.rc && hipStreamSynchronize(c.stream) != hipSuccess) fail(W2RAP_E_HIP, "staged gather of the solid k-mers failed");
                    a_hi = t_hi; a_lo = t_lo; a_cc = t_cc; a_cs = t_cs; a_cn = t_cn;
                }
                if (!X.rc) check(w2rap_step2_dict_append(h, a_hi, a_lo, a_cc, Y.S, a_cs, a_cn, Y.nchunks));
            }
        }
        if (!X.rc && c.stream2 && hipStreamSynchronize(c.stream2) != hipSuccess) fail(W2RAP_E_HIP, "gather of the solid k-mers failed");
        if (bar.wait()) return;                                           // all copies out of the owners' arrays are complete
        check(w2rap_step2_dict_end(h, M_total, D_total, hist));
        for (void* q : X.staged_blocks) c.release(q);
        X.staged_blocks.clear();
        // ---- F: replicated graph, local pathing
        if (!X.rc) check(w2rap_step2_build_graph(h, p->edge_order_hint));
        if (!X.rc && !(p->flags & W2RAP_F_GRAPH_ONLY)) check(w2rap_step2_path_reads(h));
        if (!X.rc) check(w2rap_step2_fetch(h, &X.out));
    };
    std::vector<std::thread> th;
    for (unsigned r = 1; r < world; ++r) th.emplace_back(body, r);
    body(0);
    for (auto& t : th) t.join();
    int rc = 0; std::string msg;
    for (auto& x : R) if (x.rc && !rc) { rc = x.rc; msg = "rank " + std::to_string(&x - R.data()) + " (device " + std::to_string(x.dev) + "): " + x.err; }
    if (!rc) {
        // every rank holds the same graph; the paths of the shards follow each other in read order
        for (unsigned r = 1; r < world && !rc; ++r)
            if (R[r].out.n_edge_objs != R[0].out.n_edge_objs || R[r].out.n_vertices != R[0].out.n_vertices || R[r].out.n_unipaths != R[0].out.n_unipaths) {
                rc = W2RAP_E_GRAPH; msg = "the replicated graphs differ between ranks";
            }
    }
    if (!rc && (p->flags & W2RAP_F_GRAPH_ONLY)) {             // pPaths == nullptr: the graph and the job's statistics, no paths
        *out = R[0].out;
        std::memset(&R[0].out, 0, sizeof(R[0].out));
        out->n_kmer_instances = M_total; out->n_kmers_distinct = D_total; out->n_kmers_solid = S_total;
        for (int i = 0; i < 101; ++i) out->hist[i] = hist[i];
        for (unsigned r = 1; r < world; ++r) { out->ms_count = std::max(out->ms_count, R[r].out.ms_count); out->ms_graph = std::max(out->ms_graph, R[r].out.ms_graph); }
        if (p->freqs_path) rc = write_freqs(p->freqs_path, out->hist, msg);
    } else if (!rc) {
        *out = R[0].out;
        std::memset(&R[0].out, 0, sizeof(R[0].out));
        uint64_t total = 0;
        for (auto& x : R) total += (&x == &R[0] ? out->n_paths : x.out.n_paths) ? (&x == &R[0] ? out->path_off[out->n_paths] : x.out.path_off[x.out.n_paths]) : 0;
        int32_t* po = (int32_t*)host_result_alloc((n ? n : 1) * sizeof(int32_t));
        uint64_t* pf = (uint64_t*)host_result_alloc((n + 1) * sizeof(uint64_t));
        int32_t* pe = (int32_t*)host_result_alloc((total ? total : 1) * sizeof(int32_t));
        if (!po || !pf || !pe) { rc = W2RAP_E_HIP; msg = "out of host memory"; std::free(po); std::free(pf); std::free(pe); }
        else {
            uint64_t at = 0, eat = 0;
            for (unsigned r = 0; r < world; ++r) {
                const w2rap_step2_out& o = r ? R[r].out : *out;
                const uint64_t m = o.n_paths;
                if (m) {
                    std::memcpy(po + at, o.path_offset, m * sizeof(int32_t));
                    for (uint64_t i = 0; i < m; ++i) pf[at + i] = o.path_off[i] + eat;
                    std::memcpy(pe + eat, o.path_edges, o.path_off[m] * sizeof(int32_t));
                    at += m; eat += o.path_off[m];
                }
            }
            pf[n] = eat;
            std::free(out->path_offset); std::free(out->path_off); std::free(out->path_edges);
            out->path_offset = po; out->path_off = pf; out->path_edges = pe; out->n_paths = n;
            out->n_kmer_instances = M_total; out->n_kmers_distinct = D_total; out->n_kmers_solid = S_total;
            for (int i = 0; i < 101; ++i) out->hist[i] = hist[i];
            for (unsigned r = 1; r < world; ++r) {
                out->n_reads_pathed += R[r].out.n_reads_pathed; out->n_reads_multipathed += R[r].out.n_reads_multipathed;
                out->ms_count = std::max(out->ms_count, R[r].out.ms_count); out->ms_graph = std::max(out->ms_graph, R[r].out.ms_graph);
                out->ms_path = std::max(out->ms_path, R[r].out.ms_path);
            }
            if (p->freqs_path) rc = write_freqs(p->freqs_path, out->hist, msg);
        }
    }
    // every rank's queued work first (after a failure copies out of ANOTHER rank's buffers may still be in flight), then the contexts
    for (unsigned r = 0; r < world; ++r) {
        (void)hipSetDevice(R[r].dev);
        if (R[r].copy_stream) { (void)hipStreamSynchronize(R[r].copy_stream); (void)hipStreamDestroy(R[r].copy_stream); }
        if (rc) (void)hipDeviceSynchronize();
    }
    for (unsigned r = 0; r < world; ++r) {
        (void)hipSetDevice(R[r].dev);
        bounce[r].destroy();
        w2rap_step2_free(&R[r].out);
        if (rc) w2rap_step2_destroy(R[r].h); else w2rap_step2_release(R[r].h);
    }
    if (rc) { w2rap_step2_free(out); set_err(err, errlen, msg); }
    return rc;
}

}  // namespace

extern "C" int w2rap_step2_last_peer_mode(void) { return g_last_peer_mode.load(); }

extern "C" int w2rap_step2_run(const w2rap_reads* reads, const w2rap_step2_params* p, w2rap_step2_out* out, char* err, size_t errlen) {
    if (!reads || !p || !out) { set_err(err, errlen, "null argument"); return W2RAP_E_ARG; }
    if (p->K != 60) { set_err(err, errlen, "K must be 60 (BuildReadQGraph.cc:51)"); return W2RAP_E_ARG; }
    std::memset(out, 0, sizeof(*out));
    const int world = p->n_gpus > 1 ? p->n_gpus : 1;
    if (world > 64) { set_err(err, errlen, "n_gpus: at most 64"); return W2RAP_E_ARG; }
    if (world == 1) return run_single(reads, p, p->devices ? p->devices[0] : p->device, out, err, errlen);
    if (p->n_passes > 64) { set_err(err, errlen, "n_passes: at most 64"); return W2RAP_E_ARG; }
    std::vector<int> devs(world);
    for (int r = 0; r < world; ++r) devs[r] = p->devices ? p->devices[r] : p->device + r;
    return run_multi(reads, p, (unsigned)world, devs.data(), out, err, errlen);
}
