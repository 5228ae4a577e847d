// step2_capi.hip -- the C ABI of libw2rap_step2.so (include/w2rap_step2.h).
// Host-side plumbing only: context, read upload, phase sequencing, result download.
// There is no CPU implementation of any phase here; if the GPU is missing the entry
// points fail with W2RAP_E_NO_DEVICE.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include "ctx.h"

using namespace w2;


namespace {

void set_err(char* err, size_t errlen, const std::string& m) {
    if (err && errlen) { std::snprintf(err, errlen, "%s", m.c_str()); }
}


// object o's packed bytes: thread per output byte, object found by binary search
__global__ void __launch_bounds__(256) k_obj_len(uint64_t NO, const uint32_t* __restrict__ obj_edge, const uint32_t* __restrict__ edge_nk,
                                                  uint32_t* __restrict__ len, uint32_t* __restrict__ nbytes) {
    uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= NO) return;
    uint32_t l = edge_nk[obj_edge[o] >> 1] + (K - 1);
    len[o] = l; nbytes[o] = (l + 3) >> 2;
}
__global__ void __launch_bounds__(256) k_pack_objs(uint64_t total_bytes, uint64_t NO, const uint64_t* __restrict__ byte_off,
                                                    const uint32_t* __restrict__ obj_edge, const uint32_t* __restrict__ edge_nk,
                                                    const uint64_t* __restrict__ edge_off, const uint8_t* __restrict__ codes,
                                                    uint8_t* __restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total_bytes) return;
    uint64_t lo = 0, hi = NO;                       // largest o with byte_off[o] <= i
    while (hi - lo > 1) { uint64_t m = (lo + hi) >> 1; if (byte_off[m] <= i) lo = m; else hi = m; }
    uint32_t oe = obj_edge[lo], e = oe >> 1; bool rc = oe & 1;
    uint32_t len = edge_nk[e] + (K - 1);
    uint64_t eo = edge_off[e];
    uint32_t t0 = (uint32_t)(i - byte_off[lo]) * 4;
    unsigned v = 0;
    for (unsigned j = 0; j < 4; ++j) {
        uint32_t t = t0 + j;
        if (t < len) {
            unsigned b = rc ? 3u - codes[eo + (len - 1 - t)] : codes[eo + t];
            v |= b << (2 * j);
        }
    }
    out[i] = (uint8_t)v;
}


}  // namespace
namespace w2 {
// (also used by step1_ingest.hip, which installs its output as the context's reads)
void drop_reads(Ctx& c) {
    (void)quals_wait(c);                               // (a late quality upload still writing one of these blocks)
    if (c.copy_stream) (void)hipStreamSynchronize(c.copy_stream);
    c.d_qmask = nullptr; c.qmask_min_qual = -1; c.quals_absent = false;
    for (void* p : c.owned_reads) c.park(p);           // blocks from Ctx::alloc are parked for reuse, foreign ones freed
    c.owned_reads.clear();
    c.d_bases = nullptr; c.d_boff = nullptr; c.d_len = nullptr; c.d_quals = nullptr; c.d_qoff = nullptr; c.n = 0;
}
void drop_results(Ctx& c) {
    if (c.stream) (void)hipStreamSynchronize(c.stream);
    if (c.stream2) (void)hipStreamSynchronize(c.stream2);
    shard_free(c);
    c.free_all();
    c.d_good = nullptr; c.d_bcount = nullptr; c.d_bbase = nullptr; c.d_recs = nullptr; c.d_shi = c.d_slo = nullptr; c.d_scc = nullptr;
    c.d_table = nullptr; c.d_filter32 = nullptr; c.f32words = 0; c.d_sctx = nullptr; c.d_nbr = nullptr; c.d_srec = nullptr; c.d_index = nullptr; c.index_cap = 0; c.d_xindex = nullptr; c.xindex_cap = 0; c.d_unres = nullptr; c.fused_prune = false; c.unfused_chunks.clear();
    c.d_chunk_start = nullptr; c.d_chunk_cnt = nullptr; c.nchunks = 0;
    for (unsigned k = 0; k < c.cs_ns; ++k) (void)hipEventDestroy(c.cs_ev[k]);
    c.cs_ns = 0; c.cs_planned = 0; c.cs_cnt = nullptr; c.cs_off = nullptr; c.cs_defer = nullptr; c.pass = 0; c.npass = 1; c.pass_cnt = nullptr;
    c.g_hi = c.g_lo = nullptr; c.g_cc = nullptr; c.g_cstart = nullptr; c.g_ccnt = nullptr; c.g_open = false; c.g_n = c.g_nc = 0;
    c.d_edge_nk = nullptr; c.d_edge_off = nullptr; c.d_edge_codes = nullptr; c.d_edge_bits = nullptr; c.d_fwdX = c.d_revX = nullptr; c.d_obj_edge = nullptr;
    c.d_otab = nullptr; c.d_left = c.d_right = nullptr; c.d_from_off = c.d_to_off = nullptr; c.d_from_v = c.d_from_e = c.d_to_v = c.d_to_e = nullptr;
    c.d_path_offset = nullptr; c.d_path_off = nullptr; c.d_path_edges = nullptr;
    c.quality_done = c.counted = c.graphed = c.pathed_done = false; c.table_built = false;
    c.M = c.D = c.S = c.E = c.NO = c.NV = 0; c.path_total = 0; c.n_pathed = c.n_multipathed = 0;
}
}  // namespace w2
namespace {

// a host array -> device block from the context's pool (kept with the reads), `pad` zeroed elements behind it; big arrays travel
// through the pinned staging pump
template <class T>
int up(Ctx& c, const T** dev, const T* host, uint64_t n, uint64_t pad = 0) {
    T* p = c.alloc<T>(n + pad + 1, false);
    if (!p) return W2RAP_E_HIP;
    c.owned_reads.push_back(p);
    W2_HIP(hipMemsetAsync(p + n, 0, (pad + 1) * sizeof(T), c.stream));
    if (n) W2_TRY(pump_upload(c, p, host, n * sizeof(T)));
    *dev = (const T*)p;
    return 0;
}

}  // namespace

extern "C" {

int w2rap_step2_abi_version(void) { return W2RAP_STEP2_ABI_VERSION; }

int w2rap_step2_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

w2rap_step2_ctx* w2rap_step2_create(int device, char* err, size_t errlen) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { set_err(err, errlen, "no HIP device (libw2rap_step2 has no CPU fallback)"); return nullptr; }
    if (device < 0 || device >= n) { set_err(err, errlen, "device ordinal out of range"); return nullptr; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { set_err(err, errlen, "hipGetDeviceProperties failed"); return nullptr; }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_err(err, errlen, std::string("device is ") + prop.gcnArchName + "; this library is built for gfx950 (MI355X) only");
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) { set_err(err, errlen, "hipSetDevice failed"); return nullptr; }
    auto* h = new w2rap_step2_ctx;
    h->c.device = device;
    h->c.sm_count = prop.multiProcessorCount;
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);        // numerically lower = higher priority
    // (round 4, measured: the side stream above the main one costs the partition 3.5 ms -- K2 takes K1's issue slots --, level with it: no change)
    if (hipStreamCreateWithPriority(&h->c.stream, hipStreamNonBlocking, prio_hi) != hipSuccess ||
        hipStreamCreateWithPriority(&h->c.stream2, hipStreamNonBlocking, prio_lo) != hipSuccess ||
        hipHostMalloc((void**)&h->c.h_pinned, 64 * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) {
        set_err(err, errlen, "hipStreamCreate / hipHostMalloc failed"); delete h; return nullptr;
    }
    return h;
}

void w2rap_step2_destroy(w2rap_step2_ctx* h) {
    if (!h) return;
    (void)hipSetDevice(h->c.device);
    (void)hipStreamSynchronize(h->c.stream);
    if (h->c.stream2) (void)hipStreamSynchronize(h->c.stream2);
    drop_results(h->c);
    drop_reads(h->c);
    h->c.trim();
    pump_free(h->c);
    (void)hipStreamDestroy(h->c.stream);
    if (h->c.g_copied) (void)hipEventDestroy(h->c.g_copied);
    if (h->c.stream2) (void)hipStreamDestroy(h->c.stream2);
    if (h->c.h_pinned) (void)hipHostFree(h->c.h_pinned);
    delete h;
}

const char* w2rap_step2_last_error(const w2rap_step2_ctx* h) { return h ? h->c.err.c_str() : "null context"; }
void* w2rap_step2_stream(w2rap_step2_ctx* h) { return h ? (void*)h->c.stream : nullptr; }

int w2rap_step2_set_reads(w2rap_step2_ctx* h, const w2rap_reads* r) {
    if (!h || !r) return W2RAP_E_ARG;
    Ctx& c = h->c;
    W2_HIP(hipSetDevice(c.device));
    drop_results(c);
    drop_reads(c);
    const uint64_t n = r->n_reads;
    if (n && (!r->bases_packed || !r->base_byte_off || !r->read_len)) { c.err = "set_reads: null base arrays"; return W2RAP_E_ARG; }
    const bool raw = r->quals && r->qual_off, pq = r->pq && r->pq_off;
    if (n && raw == pq) { c.err = "set_reads: give exactly one of (quals, qual_off) and (pq, pq_off)"; return W2RAP_E_ARG; }
    if (r->mem == W2RAP_MEM_DEVICE) {
        // The caller's arrays may still be being written by kernels on ITS streams (e.g. a torch generator on the default stream), and
        // the library's streams are non-blocking: wait once for the whole device before the first of our kernels reads them.
        W2_HIP(hipDeviceSynchronize());
        c.d_bases = r->bases_packed; c.d_boff = r->base_byte_off; c.d_len = r->read_len;
        if (raw || !n) { c.d_quals = r->quals; c.d_qoff = r->qual_off; }
        else {
            // qual_off = prefix sum of read_len, computed on the device
            uint64_t* qoff = nullptr; W2_HIP(hipMalloc((void**)&qoff, (n + 1) * 8)); c.owned_reads.push_back(qoff);
            W2_TRY(exclusive_scan_u32_to_u64(c, c.d_len, qoff, n));
            uint64_t total = 0;
            W2_HIP(hipMemcpy(&total, qoff + n, 8, hipMemcpyDeviceToHost));
            uint8_t* q = nullptr; W2_HIP(hipMalloc((void**)&q, total ? total : 1)); c.owned_reads.push_back(q);
            c.n = n;
            W2_TRY(decode_pq(c, r->pq, r->pq_off, q, qoff));
            c.d_quals = q; c.d_qoff = qoff;
        }
    } else if (r->mem == W2RAP_MEM_HOST) {
        // host arrays are checked before a kernel indexes by them: every read's packed bytes and qualities must be what its length says
        // (device arrays are the caller's own kernels' output and are taken as they are)
        {   // (a sweep over three arrays of n entries: split over the worker threads, the first offending read is reported)
            const size_t nblk = (size_t)std::min<uint64_t>(64, (n + 65535) / 65536);
            std::vector<uint64_t> bad(nblk ? nblk : 1, ~0ull); std::vector<int> why(nblk ? nblk : 1, 0);
            host_parallel_for(nblk, [&](size_t b) {
                const uint64_t lo = n * b / nblk, hi = n * (b + 1) / nblk;
                for (uint64_t i = lo; i < hi; ++i) {
                    const uint64_t nb = r->base_byte_off[i + 1] - r->base_byte_off[i];
                    int w = 0;
                    if (r->base_byte_off[i + 1] < r->base_byte_off[i] || nb != ((uint64_t)r->read_len[i] + 3) / 4) w = 1;
                    else if (raw && (r->qual_off[i + 1] < r->qual_off[i] || r->qual_off[i + 1] - r->qual_off[i] != r->read_len[i])) w = 2;
                    else if (!raw && r->pq_off[i + 1] <= r->pq_off[i]) w = 3;
                    else if (r->read_len[i] > 65535u) w = 4;
                    if (w) { bad[b] = i; why[b] = w; return; }
                }
            });
            for (size_t b = 0; b < nblk; ++b) if (bad[b] != ~0ull) {
                const std::string at = " (read " + std::to_string(bad[b]);
                if (why[b] == 4) { c.err = "a read of more than 65,535 bases" + at + "): good lengths are 16-bit words"; return W2RAP_E_LIMIT; }
                c.err = why[b] == 1 ? "set_reads: base_byte_off does not match read_len" + at + ")" : why[b] == 2 ? "set_reads: qual_off does not match read_len" + at + ")"
                                    : "set_reads: pq_off is not ascending" + at + "; every PQVec ends with a 0 byte)";
                return W2RAP_E_ARG;
            }
        }
        if (n && (r->base_byte_off[0] != 0 || (raw ? r->qual_off[0] : r->pq_off[0]) != 0)) { c.err = "set_reads: offsets must start at 0"; return W2RAP_E_ARG; }
        const uint64_t nbytes = n ? r->base_byte_off[n] : 0;
        // The sweep above has established that base_byte_off and qual_off ARE the prefix sums of ceil(len / 4) and len: they are computed on
        // the device from the lengths instead of travelling (16 B per read: 0.8 GB of the 10.4 GB of 50 M PE150 reads).
        auto derived = [&](const uint64_t** dev, bool packed) -> int {
            uint64_t* p = c.alloc<uint64_t>(n + 2, false);
            if (!p) return W2RAP_E_HIP;
            c.owned_reads.push_back(p);
            W2_TRY(packed ? exclusive_scan_packed_bytes(c, c.d_len, p, n) : exclusive_scan_u32_to_u64(c, c.d_len, p, n));
            *dev = p;
            return 0;
        };
        const uint64_t nq_raw = (raw && n) ? r->qual_off[n] : 0;
        const bool late_quals = raw && c.hint_min_qual >= 0 && nq_raw >= (64ull << 20) && !getenv("W2RAP_NO_UPLOAD_OVERLAP");
        if (late_quals) W2_TRY(quality_mask_begin(c, r->quals, nq_raw, (uint32_t)c.hint_min_qual));      // made on the host while the bases travel
        struct MaskGuard { Ctx& c; ~MaskGuard() { quality_mask_cancel(c); } } mask_guard{c};               // (an early return joins its threads)
        const bool tr = getenv("W2RAP_TRACE") != nullptr;
        auto tnow = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double t_a = tnow();
        W2_TRY(up(c, &c.d_len, r->read_len, n));
        W2_TRY(up(c, &c.d_bases, r->bases_packed, nbytes, 32));
        W2_TRY(derived(&c.d_boff, true));
        const double t_b = tnow();
        if (raw || !n) {
            const uint64_t nq = n ? r->qual_off[n] : 0;
            if (late_quals) {
                // w2rap_step2_run: what the counting needs of the qualities is one bit per base -- made on the host side of the pump,
                // 1/8 of the bytes --; the raw qualities (read by the path extension's scores only) follow on a copy stream, fed by a
                // host thread, while the counting and the graph phase run (quals_wait: before read pathing)
                uint8_t* q = c.alloc<uint8_t>(nq + 33, false);
                uint32_t* m = c.alloc<uint32_t>((nq + 31) / 32 + 17, false);
                if (!q || !m) return W2RAP_E_HIP;
                c.owned_reads.push_back(q); c.owned_reads.push_back(m);
                W2_HIP(hipMemsetAsync(q + nq, 0, 33, c.stream));
                W2_HIP(hipMemsetAsync(m + nq / 32, 0, ((nq + 31) / 32 + 17 - nq / 32) * 4, c.stream));        // (the last, partial word and the pad; the mask's bytes come behind this on the same stream)
                W2_TRY(quality_mask_upload(c, m));
                c.d_qmask = m; c.qmask_min_qual = c.hint_min_qual; c.d_quals = q;
                W2_TRY(derived(&c.d_qoff, false));
                const double t_c = tnow();
                W2_HIP(hipStreamSynchronize(c.stream));               // the bases and the mask are up: the qualities get the link to themselves
                if (tr) fprintf(stderr, "[w2rap] set_reads: lengths + bases queued after %.1f ms, mask made and queued after %.1f more, all up after %.1f more\n", t_b - t_a, t_c - t_b, tnow() - t_c);
                // read pathing starts on the first 60 % of the reads while the last 40 % of the qualities are still on their way
                uint64_t pre = (uint64_t)((double)n * 0.6) & ~1ull;
                if (pre >= n) pre = 0;
                if (c.hint_graph_only) c.quals_absent = true;        // pPaths == nullptr (BuildReadQGraph.cc:1300-1307): nothing will read them
                else W2_TRY(quals_upload_begin(c, q, r->quals, nq, pre ? r->qual_off[pre] : 0, pre));
            } else {
                W2_TRY(up(c, &c.d_quals, r->quals, nq, 32));
                W2_TRY(derived(&c.d_qoff, false));
            }
        } else {
            const uint8_t* d_pq = nullptr; const uint64_t* d_pqoff = nullptr;
            W2_TRY(up(c, &d_pq, r->pq, r->pq_off[n], 32));
            W2_TRY(up(c, &d_pqoff, r->pq_off, n + 1));
            // qual_off = prefix sum of read_len, computed on the device; the PQVec blobs are decoded there (k_decode_pq)
            uint64_t* qoff = c.alloc<uint64_t>(n + 1, false);
            if (!qoff) return W2RAP_E_HIP;
            c.owned_reads.push_back(qoff);
            W2_TRY(exclusive_scan_u32_to_u64(c, c.d_len, qoff, n));
            uint64_t total = 0;
            W2_HIP(hipMemcpyAsync(&total, qoff + n, 8, hipMemcpyDeviceToHost, c.stream));
            W2_HIP(hipStreamSynchronize(c.stream));
            uint8_t* q = c.alloc<uint8_t>(total + 32, false);
            if (!q) return W2RAP_E_HIP;
            c.owned_reads.push_back(q);
            c.n = n;
            c.d_qoff = qoff;
            W2_TRY(decode_pq(c, d_pq, d_pqoff, q, c.d_qoff));
            c.d_quals = q;
        }
    } else { c.err = "set_reads: bad mem kind"; return W2RAP_E_ARG; }
    c.n = n;
    W2_HIP(hipStreamSynchronize(c.stream));
    return 0;
}

static void fill_stats(const Ctx& c, w2rap_step2_out* s) {
    if (!s) return;
    for (int i = 0; i < 101; ++i) s->hist[i] = c.hist[i];
    s->n_kmer_instances = c.M; s->n_kmers_distinct = c.D; s->n_kmers_solid = c.S;
    s->n_reads_pathed = c.n_pathed; s->n_reads_multipathed = c.n_multipathed;
    s->ms_count = c.ms_count; s->ms_graph = c.ms_graph; s->ms_path = c.ms_path;
    s->K = 60;
}

int w2rap_step2_counts(w2rap_step2_ctx* h, uint64_t out[8]) {
    if (!h || !out) return W2RAP_E_ARG;
    const Ctx& c = h->c;
    out[0] = c.M; out[1] = c.D; out[2] = c.S; out[3] = c.graphed ? c.E : 0; out[4] = c.graphed ? c.NO : 0; out[5] = c.graphed ? c.NV : 0;
    out[6] = c.pathed_done ? c.n_pathed : 0; out[7] = c.pathed_done ? c.path_total : 0;
    return 0;
}

int w2rap_step2_count_kmers(w2rap_step2_ctx* h, uint32_t min_qual, uint32_t min_freq, w2rap_step2_out* stats) {
    if (!h) return W2RAP_E_ARG;
    Ctx& c = h->c;
    W2_HIP(hipSetDevice(c.device));
    if (c.n && !c.d_bases) { c.err = "count_kmers called before set_reads"; return W2RAP_E_STATE; }
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now();
    drop_results(c);
    if (getenv("W2RAP_TRACE")) fprintf(stderr, "[w2rap] drop_results %.1f ms\n", (now() - t0) * 1e3);
    Timer t(c.stream);
    int rc = phase_count(c, min_qual, min_freq);
    c.ms_count = t.stop();
    c.presolve();
    if (rc) return rc;
    fill_stats(c, stats);
    return 0;
}

int w2rap_step2_count_kmers_passes(w2rap_step2_ctx* h, uint32_t min_qual, uint32_t min_freq, uint32_t n_passes, w2rap_step2_out* stats) {
    if (!h) return W2RAP_E_ARG;
    const uint32_t keep = h->c.n_passes;
    h->c.n_passes = n_passes;
    const int rc = w2rap_step2_count_kmers(h, min_qual, min_freq, stats);
    h->c.n_passes = keep;
    return rc;
}

// ---- multi-GPU building blocks: the same kernels as count_kmers, split at the
// points where the host code exchanges data between ranks (RCCL all_to_all_v / all_gather_v).
int w2rap_step2_quality_windows(w2rap_step2_ctx* h, uint32_t min_qual, uint64_t* n_kmers) {
    if (!h) return W2RAP_E_ARG;
    Ctx& c = h->c;
    W2_HIP(hipSetDevice(c.device));
    if (c.n && !c.d_bases) { c.err = "quality_windows called before set_reads"; return W2RAP_E_STATE; }
    drop_results(c);
    Timer t(c.stream);
    int rc = count_quality(c, min_qual);
    c.ms_count = t.stop();
    c.presolve();
    if (!rc && n_kmers) *n_kmers = c.M;
    return rc;
}

uint32_t w2rap_step2_default_buckets(uint64_t total_kmers, uint32_t multiple_of) { return default_buckets(total_kmers, multiple_of); }
uint32_t w2rap_step2_record_bytes(void) { return REC_BYTES; }

int w2rap_step2_partition_range(w2rap_step2_ctx* h, uint32_t n_buckets, uint32_t first_bucket, uint32_t end_bucket, uint32_t n_parts,
                                uint64_t* recs_per_part, uint64_t* kmers_per_part) {
    if (!h || !n_buckets || !n_parts || end_bucket > n_buckets || first_bucket >= end_bucket || (end_bucket - first_bucket) % n_parts) return W2RAP_E_ARG;
    Ctx& c = h->c;
    W2_HIP(hipSetDevice(c.device));
    Timer t(c.stream);
    if (n_parts > 64) { c.err = "partition: more than 64 parts"; return W2RAP_E_LIMIT; }
    int rc = count_partition(c, n_buckets, kmers_per_part ? n_parts : 0, first_bucket, end_bucket);
    c.ms_count += t.stop();
    c.presolve();
    if (rc) return rc;
    if (kmers_per_part) for (uint32_t g = 0; g < n_parts; ++g) kmers_per_part[g] = c.part_kmers[g];
    if (recs_per_part) {
        const uint32_t nbl = (end_bucket - first_bucket) / n_parts;
        uint64_t prev = 0;
        for (uint32_t g = 1; g <= n_parts; ++g) {
            uint64_t b = 0;
            W2_HIP(hipMemcpy(&b, c.d_bbase + (uint64_t)g * nbl, 8, hipMemcpyDeviceToHost));
            recs_per_part[g - 1] = b - prev; prev = b;
        }
    }
    return 0;
}
int w2rap_step2_partition(w2rap_step2_ctx* h, uint32_t n_buckets, uint32_t n_parts, uint64_t* recs_per_part, uint64_t* kmers_per_part) {
    return w2rap_step2_partition_range(h, n_buckets, 0, n_buckets, n_parts, recs_per_part, kmers_per_part);
}

// the count_records call(s) that follow are hash-range pass `pass` of `n_passes` on this owner: a later pass appends to the solid k-mers,
// chunks, counters and histogram of the passes before it (MapReduceEngine.h:288-299)
int w2rap_step2_count_pass(w2rap_step2_ctx* h, uint32_t pass, uint32_t n_passes) {
    if (!h || !n_passes || pass >= n_passes) return W2RAP_E_ARG;
    Ctx& c = h->c;
    if (c.cs_planned) { c.err = "count_pass while a sliced count is pending"; return W2RAP_E_STATE; }
    if (pass && !c.pass_cnt) { c.err = "count_pass: passes go in order, each one counted before the next"; return W2RAP_E_STATE; }
    c.pass = pass; c.npass = n_passes;
    return 0;
}

int w2rap_step2_partition_buffers(w2rap_step2_ctx* h, void** d_records, void** d_bucket_counts, uint64_t* n_records) {
    if (!h) return W2RAP_E_ARG;
    Ctx& c = h->c;
    if (!c.d_bcount) { c.err = "partition_buffers before partition"; return W2RAP_E_STATE; }
    if (d_records) *d_records = c.d_recs;
    if (d_bucket_counts) *d_bucket_counts = c.d_bcount;
    if (n_records) *n_records = c.nrec;
    return 0;
}

int w2rap_step2_count_records_begin(w2rap_step2_ctx* h, uint32_t min_freq, uint32_t n_local_buckets, uint32_t n_segments, const void* d_records,
                                    const void* d_counts, uint64_t total_kmers, uint32_t n_slices, int deferred) {
    if (!h || !n_local_buckets || !n_segments || !d_counts) return W2RAP_E_ARG;
    Ctx& c = h->c;
    W2_HIP(hipSetDevice(c.device));
    return count_buckets_launch(c, min_freq, n_local_buckets, n_segments, (const uint32_t*)d_records, (const uint32_t*)d_counts, total_kmers,
                                n_slices ? n_slices : 1, deferred != 0);
}

int w2rap_step2_count_records_slices(w2rap_step2_ctx* h) { return h ? (int)h->c.cs_planned : 0; }

int w2rap_step2_count_records_bounds(w2rap_step2_ctx* h, uint32_t k, uint32_t* first_bucket, uint32_t* end_bucket) {
    if (!h || !first_bucket || !end_bucket) return W2RAP_E_ARG;
    Ctx& c = h->c;
    if (!c.cs_planned || k >= c.cs_planned) { c.err = "count_records_bounds: no such slice"; return W2RAP_E_ARG; }
    count_slice_bounds(c, k, first_bucket, end_bucket);
    return 0;
}

int w2rap_step2_count_records_launch(w2rap_step2_ctx* h, uint32_t k) {
    if (!h) return W2RAP_E_ARG;
    Ctx& c = h->c;
    W2_HIP(hipSetDevice(c.device));
    return count_buckets_launch_slice(c, k);
}

int w2rap_step2_count_records_slice(w2rap_step2_ctx* h, uint32_t k, uint64_t* n_solid, uint64_t* n_chunks) {
    if (!h) return W2RAP_E_ARG;
    Ctx& c = h->c;
    W2_HIP(hipSetDevice(c.device));
    return count_buckets_slice(c, k, n_solid, n_chunks);
}

int w2rap_step2_count_records_end(w2rap_step2_ctx* h, w2rap_step2_out* stats) {
    if (!h) return W2RAP_E_ARG;
    Ctx& c = h->c;
    W2_HIP(hipSetDevice(c.device));
    int rc = count_buckets_finish(c);
    c.presolve();
    if (rc) return rc;
    fill_stats(c, stats);
    return 0;
}

int w2rap_step2_count_records(w2rap_step2_ctx* h, uint32_t min_freq, uint32_t n_local_buckets, uint32_t n_segments, const void* d_records,
                              const void* d_counts, uint64_t total_kmers, w2rap_step2_out* stats) {
    if (!h || !n_local_buckets || !n_segments || !d_counts) return W2RAP_E_ARG;
    Ctx& c = h->c;
    W2_HIP(hipSetDevice(c.device));
    Timer t(c.stream);
    int rc = count_buckets(c, min_freq, n_local_buckets, n_segments, (const uint32_t*)d_records, (const uint32_t*)d_counts, total_kmers);
    c.ms_count += t.stop();
    c.presolve();
    if (rc) return rc;
    fill_stats(c, stats);
    return 0;
}

int w2rap_step2_solid_buffers(w2rap_step2_ctx* h, void** d_hi, void** d_lo, void** d_cc, uint64_t* n) {
    if (!h) return W2RAP_E_ARG;
    Ctx& c = h->c;
    if (!c.d_shi) { c.err = "solid_buffers before count_records"; return W2RAP_E_STATE; }
    if (d_hi) *d_hi = c.d_shi;
    if (d_lo) *d_lo = c.d_slo;
    if (d_cc) *d_cc = c.d_scc;
    if (n) *n = c.S;
    return 0;
}

int w2rap_step2_chunk_buffers(w2rap_step2_ctx* h, void** d_start, void** d_count, uint64_t* n) {
    if (!h) return W2RAP_E_ARG;
    Ctx& c = h->c;
    if (!c.d_shi) { c.err = "chunk_buffers before count_records"; return W2RAP_E_STATE; }
    if (d_start) *d_start = c.d_chunk_start;
    if (d_count) *d_count = c.d_chunk_cnt;
    if (n) *n = c.nchunks;
    return 0;
}

int w2rap_step2_dict_begin(w2rap_step2_ctx* h, uint64_t kmer_capacity, uint64_t chunk_capacity) {
    if (!h) return W2RAP_E_ARG;
    Ctx& c = h->c;
    W2_HIP(hipSetDevice(c.device));
    return dict_begin(c, kmer_capacity, chunk_capacity);
}

int w2rap_step2_dict_append(w2rap_step2_ctx* h, const void* d_hi, const void* d_lo, const void* d_cc, uint64_t n,
                            const void* d_chunk_start, const void* d_chunk_count, uint64_t n_chunks) {
    if (!h || (n && (!d_hi || !d_lo || !d_cc)) || (n_chunks && (!d_chunk_start || !d_chunk_count))) return W2RAP_E_ARG;
    Ctx& c = h->c;
    W2_HIP(hipSetDevice(c.device));
    return dict_append(c, (const uint64_t*)d_hi, (const uint64_t*)d_lo, (const uint32_t*)d_cc, n, (const uint64_t*)d_chunk_start,
                       (const uint32_t*)d_chunk_count, n_chunks);
}

// the same for a SLICE of an owner's arrays handed over in place (peer memory): the chunk starts count from k-mer `chunk_bias` of the owner's array
int w2rap_step2_dict_append_slice(w2rap_step2_ctx* h, const void* d_hi, const void* d_lo, const void* d_cc, uint64_t n,
                                  const void* d_chunk_start, const void* d_chunk_count, uint64_t n_chunks, uint64_t chunk_bias) {
    if (!h || (n && (!d_hi || !d_lo || !d_cc)) || (n_chunks && (!d_chunk_start || !d_chunk_count))) return W2RAP_E_ARG;
    Ctx& c = h->c;
    W2_HIP(hipSetDevice(c.device));
    return dict_append(c, (const uint64_t*)d_hi, (const uint64_t*)d_lo, (const uint32_t*)d_cc, n, (const uint64_t*)d_chunk_start,
                       (const uint32_t*)d_chunk_count, n_chunks, chunk_bias);
}

int w2rap_step2_dict_end(w2rap_step2_ctx* h, uint64_t M, uint64_t D, const uint64_t* hist101) {
    if (!h) return W2RAP_E_ARG;
    Ctx& c = h->c;
    W2_HIP(hipSetDevice(c.device));
    Timer t(c.stream);
    c.M = M; c.D = D;
    if (hist101) for (int i = 0; i < 101; ++i) c.hist[i] = hist101[i];
    int rc = dict_end(c);
    c.ms_count += t.stop();
    c.presolve();
    return rc;
}

int w2rap_step2_dict_abort(w2rap_step2_ctx* h) {
    if (!h) return W2RAP_E_ARG;
    (void)hipSetDevice(h->c.device);
    dict_abort(h->c);
    return 0;
}

int w2rap_step2_set_solid_chunked(w2rap_step2_ctx* h, const void* d_hi, const void* d_lo, const void* d_cc, uint64_t n, uint64_t M, uint64_t D,
                                  const uint64_t* hist101, const void* d_chunk_start, const void* d_chunk_count, uint64_t n_chunks) {
    if (!h || (n && (!d_hi || !d_lo || !d_cc)) || (n_chunks && (!d_chunk_start || !d_chunk_count))) return W2RAP_E_ARG;
    Ctx& c = h->c;
    W2_HIP(hipSetDevice(c.device));
    if (c.cs_planned) { c.err = "set_solid while a sliced count is pending (count_records_end first)"; return W2RAP_E_STATE; }
    W2_TRY(dict_begin(c, n, n_chunks));
    W2_TRY(dict_append(c, (const uint64_t*)d_hi, (const uint64_t*)d_lo, (const uint32_t*)d_cc, n, (const uint64_t*)d_chunk_start,
                       (const uint32_t*)d_chunk_count, n_chunks));
    W2_HIP(hipStreamSynchronize(c.stream2));             // the caller's arrays are free again when set_solid returns
    return w2rap_step2_dict_end(h, M, D, hist101);
}

int w2rap_step2_set_solid(w2rap_step2_ctx* h, const void* d_hi, const void* d_lo, const void* d_cc, uint64_t n, uint64_t M, uint64_t D,
                          const uint64_t* hist101) {
    return w2rap_step2_set_solid_chunked(h, d_hi, d_lo, d_cc, n, M, D, hist101, nullptr, nullptr, 0);
}

int w2rap_step2_build_graph(w2rap_step2_ctx* h, const w2rap_edge_hint* hint) {
    if (!h) return W2RAP_E_ARG;
    Ctx& c = h->c;
    W2_HIP(hipSetDevice(c.device));
    if (c.graphed) { c.err = "build_graph called twice; call count_kmers again"; return W2RAP_E_STATE; }
    if (hint) {                                       // the hint's arrays are read by kernels: their offsets must be what the lengths say
        if (hint->n_edges && (!hint->packed || !hint->byte_off || !hint->len)) { c.err = "edge_order_hint: null array"; return W2RAP_E_HINT; }
        for (uint64_t e = 0; e < hint->n_edges; ++e)
            if (hint->byte_off[e + 1] < hint->byte_off[e] || hint->byte_off[e + 1] - hint->byte_off[e] != ((uint64_t)hint->len[e] + 3) / 4) {
                c.err = "edge_order_hint: byte_off does not match len"; return W2RAP_E_HINT;
            }
    }
    Timer t(c.stream);
    int rc = phase_graph(c, hint);
    c.ms_graph = t.stop();
    c.presolve();
    return rc;
}

int w2rap_step2_path_reads(w2rap_step2_ctx* h) {
    if (!h) return W2RAP_E_ARG;
    Ctx& c = h->c;
    W2_HIP(hipSetDevice(c.device));
    if (c.pathed_done) { c.err = "path_reads called twice"; return W2RAP_E_STATE; }
    if (c.quals_absent) { c.err = "path_reads: the reads were installed by a graph-only call, their qualities were not uploaded"; return W2RAP_E_STATE; }
    Timer t(c.stream);
    int rc = phase_path(c);
    c.ms_path = t.stop();
    c.presolve();
    return rc;
}

// plain device-to-device copy, 16 B per lane, four independent loads in flight per lane and iteration: the box's own streaming rate, quoted
// beside the 8 TB/s spec (MI355X_MICROARCH.md reaches ~6.3 TB/s this way).  Round 4's form (one load per iteration, 32 blocks
// per CU) measured 4.7 TB/s; the guide's grid rule (8 blocks per CU, grid-stride) and the unrolling are what was missing.  The best of three
// grid sizes is reported (a measurement aid: the rate a streaming kernel CAN reach here).
__global__ void __launch_bounds__(256) k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, uint64_t n16) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}
// two more forms of the same copy (round 6: 5.1 TB/s with the form above against the guide's 6.29): non-temporal loads and stores (the
// data is touched once: no point in keeping it in the L2 / MALL), and a CONTIGUOUS stretch per block (a block streams 64 KiB pieces of
// its own region: fewer DRAM pages open at a time than with the grid-stride interleave)
__global__ void __launch_bounds__(256) k_copy16_nt(const uint4* __restrict__ src, uint4* __restrict__ dst, uint64_t n16) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    const v4u* s4 = reinterpret_cast<const v4u*>(src); v4u* d4 = reinterpret_cast<v4u*>(dst);
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const v4u a = __builtin_nontemporal_load(s4 + i), b = __builtin_nontemporal_load(s4 + i + stride), c = __builtin_nontemporal_load(s4 + i + 2 * stride),
                  d = __builtin_nontemporal_load(s4 + i + 3 * stride);
        __builtin_nontemporal_store(a, d4 + i); __builtin_nontemporal_store(b, d4 + i + stride); __builtin_nontemporal_store(c, d4 + i + 2 * stride);
        __builtin_nontemporal_store(d, d4 + i + 3 * stride);
    }
    for (; i < n16; i += stride) d4[i] = s4[i];
}
__global__ void __launch_bounds__(256) k_copy16_blk(const uint4* __restrict__ src, uint4* __restrict__ dst, uint64_t n16) {
    const uint64_t per = (n16 + gridDim.x - 1) / gridDim.x, lo = per * blockIdx.x, hi = lo + per < n16 ? lo + per : n16;
    uint64_t i = lo + threadIdx.x;
    for (; i + 3 * 256 < hi; i += 4 * 256) {
        const uint4 a = src[i], b = src[i + 256], c = src[i + 512], d = src[i + 768];
        dst[i] = a; dst[i + 256] = b; dst[i + 512] = c; dst[i + 768] = d;
    }
    for (; i < hi; i += 256) dst[i] = src[i];
}
}  // extern "C"
namespace w2 {
// the copy kernel on a stream of the caller's choice (both pointers on the current device)
int device_copy_async(Ctx& c, void* dst, const void* src, uint64_t nbytes, hipStream_t st) {
    if (!nbytes) return 0;
    const bool aligned = ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15u) == 0;
    const uint64_t n16 = aligned ? nbytes / 16 : 0, tail = nbytes - n16 * 16;
    if (n16) {
        const unsigned grid = (unsigned)std::min<uint64_t>((n16 + 255) / 256, (uint64_t)c.sm_count * 16);
        hipLaunchKernelGGL(k_copy16_blk, dim3(grid), dim3(256), 0, st, static_cast<const uint4*>(src), static_cast<uint4*>(dst), n16);
    }
    if (tail) W2_HIP(hipMemcpyAsync(static_cast<uint8_t*>(dst) + n16 * 16, static_cast<const uint8_t*>(src) + n16 * 16, tail, hipMemcpyDeviceToDevice, st));
    return 0;
}
}  // namespace w2
extern "C" {
static int g_copy_best_form = 0;       // 0 grid-stride, 1 non-temporal, 2 contiguous per block: which one the last copy_bench found fastest
extern "C" int w2rap_step2_copy_bench_form(void) { return g_copy_best_form; }
int w2rap_step2_copy_bench(w2rap_step2_ctx* h, uint64_t nbytes, uint32_t reps, double* gb_per_s) {
    if (!h || !gb_per_s || nbytes < 4096 || !reps) return W2RAP_E_ARG;
    Ctx& c = h->c;
    W2_HIP(hipSetDevice(c.device));
    const uint64_t n16 = nbytes / 16;
    uint4 *src = nullptr, *dst = nullptr;
    W2_ALLOC(src, uint4, n16); W2_ALLOC(dst, uint4, n16);
    W2_HIP(hipMemsetAsync(src, 1, n16 * 16, c.stream));
    double best = 0;
    for (int form = 0; form < 3; ++form)
        for (unsigned per_cu : {8u, 16u, 32u}) {
            const unsigned grid = (unsigned)std::min<uint64_t>((n16 + 255) / 256, (uint64_t)c.sm_count * per_cu);
            auto go = [&] {
                if (form == 0) hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(256), 0, c.stream, src, dst, n16);
                else if (form == 1) hipLaunchKernelGGL(k_copy16_nt, dim3(grid), dim3(256), 0, c.stream, src, dst, n16);
                else hipLaunchKernelGGL(k_copy16_blk, dim3(grid), dim3(256), 0, c.stream, src, dst, n16);
            };
            go();                                                             // warm-up (page tables, clocks)
            float ms = 0;
            {
                Timer t(c.stream);
                for (uint32_t r = 0; r < reps; ++r) go();
                ms = t.stop();
            }
            const double rate = 2.0 * (double)(n16 * 16) * reps / ((double)ms * 1e-3) / 1e9;         // bytes read + bytes written
            if (getenv("W2RAP_TRACE")) fprintf(stderr, "[w2rap] copy bench: form %d, %u blocks per CU: %.0f GB/s\n", form, per_cu, rate);
            if (rate > best) { best = rate; g_copy_best_form = form; }
        }
    W2_HIP(hipGetLastError());
    c.release(src); c.release(dst);
    *gb_per_s = best;
    return 0;
}

// A plain device-to-device copy by the library's own copy kernel, complete when it returns.  What the host layers use for the part of an
// exchange that stays on the rank (its own buckets' records, a world-1 run's "exchanges"): hipMemcpyAsync device-to-device goes through the
// SDMA engines at ~180 GB/s on this platform and an RCCL send to oneself at ~30 GB/s, the copy kernel moves 2.5 TB/s.
int w2rap_step2_device_copy(w2rap_step2_ctx* h, void* dst, const void* src, uint64_t nbytes) {
    if (!h || (nbytes && (!dst || !src))) return W2RAP_E_ARG;
    Ctx& c = h->c;
    W2_HIP(hipSetDevice(c.device));
    if (!nbytes) return 0;
    if (!c.copy_stream) W2_HIP(hipStreamCreateWithFlags(&c.copy_stream, hipStreamNonBlocking));
    W2_TRY(device_copy_async(c, dst, src, nbytes, c.copy_stream));
    W2_HIP(hipStreamSynchronize(c.copy_stream));
    W2_HIP(hipGetLastError());
    return 0;
}

// give the recycled device blocks of finished runs back to the driver
int w2rap_step2_trim(w2rap_step2_ctx* h) {
    if (!h) return W2RAP_E_ARG;
    (void)hipSetDevice(h->c.device);
    h->c.trim();
    return 0;
}

int w2rap_step2_set_profiling(w2rap_step2_ctx* h, int on) {
    if (!h) return W2RAP_E_ARG;
    h->c.profiling = on != 0;
    return 0;
}

// "name ms launches\n" per kernel, summed since the last reset; returns bytes needed
size_t w2rap_step2_profile(w2rap_step2_ctx* h, char* buf, size_t len, int reset) {
    if (!h) return 0;
    std::string s;
    for (auto& x : h->c.prof_sums) s += x.name + " " + std::to_string(x.ms) + " " + std::to_string(x.launches) + "\n";
    if (buf && len) std::snprintf(buf, len, "%s", s.c_str());
    if (reset) h->c.prof_sums.clear();
    return s.size() + 1;
}

int w2rap_step2_get_good_len(w2rap_step2_ctx* h, uint16_t* out) {
    if (!h || !out) return W2RAP_E_ARG;
    Ctx& c = h->c;
    if (!c.counted) { c.err = "get_good_len before count_kmers"; return W2RAP_E_STATE; }
    if (c.n) W2_HIP(hipMemcpy(out, c.d_good, c.n * 2, hipMemcpyDeviceToHost));
    return 0;
}

int w2rap_step2_get_table(w2rap_step2_ctx* h, uint64_t* hi, uint64_t* lo, uint8_t* count, uint8_t* ctx, int32_t* edge, uint32_t* off) {
    if (!h) return W2RAP_E_ARG;
    Ctx& c = h->c;
    if (!c.counted) { c.err = "get_table before count_kmers"; return W2RAP_E_STATE; }
    const uint64_t S = c.S;
    if (!S) return 0;
    if (hi) W2_HIP(hipMemcpy(hi, c.d_shi, S * 8, hipMemcpyDeviceToHost));
    if (lo) W2_HIP(hipMemcpy(lo, c.d_slo, S * 8, hipMemcpyDeviceToHost));
    if (count || (ctx && !c.counted)) {
        std::vector<uint32_t> cc(S);
        W2_HIP(hipMemcpy(cc.data(), c.d_scc, S * 4, hipMemcpyDeviceToHost));
        if (count) for (uint64_t i = 0; i < S; ++i) count[i] = (uint8_t)(cc[i] & 0xFF);
    }
    if (ctx) W2_HIP(hipMemcpy(ctx, c.d_sctx, S, hipMemcpyDeviceToHost));      // pruned context (a6)
    if (edge || off) {
        // a k-mer's (unipath, offset) is where its 60 bases lie in the edge sequences: looked up through read pathing's index (which this
        // also checks for completeness: every solid k-mer must be found)
        if (!c.graphed) { for (uint64_t i = 0; i < S; ++i) { if (edge) edge[i] = -1; if (off) off[i] = 0; } }
        else if (c.d_srec) {
            std::vector<KRec> sv(S);
            W2_HIP(hipMemcpy(sv.data(), c.d_srec, S * sizeof(KRec), hipMemcpyDeviceToHost));
            for (uint64_t i = 0; i < S; ++i) {
                if (edge) edge[i] = sv[i].kdef.x != NONE32 ? (int32_t)(sv[i].kdef.x & 0x7FFFFFFFu) : -1;
                if (off) off[i] = sv[i].kdef.y;
            }
        } else {
            int32_t* d_e = nullptr; uint32_t* d_o = nullptr;
            W2_ALLOC(d_e, int32_t, S); W2_ALLOC(d_o, uint32_t, S);
            W2_TRY(index_probe_all(c, d_e, d_o));
            W2_HIP(hipStreamSynchronize(c.stream));
            if (edge) W2_HIP(hipMemcpy(edge, d_e, S * 4, hipMemcpyDeviceToHost));
            if (off) W2_HIP(hipMemcpy(off, d_o, S * 4, hipMemcpyDeviceToHost));
            c.release(d_e); c.release(d_o);
        }
    }
    return 0;
}

int w2rap_step2_fetch(w2rap_step2_ctx* h, w2rap_step2_out* out) {
    if (!h || !out) return W2RAP_E_ARG;
    Ctx& c = h->c;
    W2_HIP(hipSetDevice(c.device));
    if (!c.graphed) { c.err = "fetch before build_graph"; return W2RAP_E_STATE; }
    std::memset(out, 0, sizeof(*out));
    fill_stats(c, out);
    hipStream_t st = c.stream;
    const uint64_t NO = c.NO, NV = c.NV, E = c.E;
    out->n_vertices = NV; out->n_edge_objs = NO; out->n_unipaths = E;
    // objects: lengths, byte offsets, packed bases
    uint32_t *d_len = nullptr, *d_nb = nullptr; uint64_t* d_boff = nullptr;
    W2_ALLOC(d_len, uint32_t, NO); W2_ALLOC(d_nb, uint32_t, NO); W2_ALLOC(d_boff, uint64_t, NO + 1);
    if (NO) hipLaunchKernelGGL(k_obj_len, dim3((unsigned)((NO + 255) / 256)), dim3(256), 0, st, NO, c.d_obj_edge, c.d_edge_nk, d_len, d_nb);
    W2_TRY(exclusive_scan_u32_to_u64(c, d_nb, d_boff, NO));
    uint64_t total = 0;
    W2_HIP(hipMemcpyAsync(&total, d_boff + NO, 8, hipMemcpyDeviceToHost, st));
    W2_HIP(hipStreamSynchronize(st));
    uint8_t* d_packed = nullptr;
    W2_ALLOC(d_packed, uint8_t, total);
    if (total) hipLaunchKernelGGL(k_pack_objs, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, total, NO, d_boff, c.d_obj_edge,
                                  c.d_edge_nk, c.d_edge_off, c.d_edge_codes, d_packed);
    W2_HIP(hipGetLastError());
    W2_TRY(dl(c, &out->edge_packed, d_packed, total));
    W2_TRY(dl(c, &out->edge_byte_off, d_boff, NO + 1));
    W2_TRY(dl(c, &out->edge_len, d_len, NO));
    W2_TRY(dl(c, &out->vleft, c.d_left, NO));
    W2_TRY(dl(c, &out->vright, c.d_right, NO));
    W2_TRY(dl(c, &out->from_off, c.d_from_off, NV + 1));
    W2_TRY(dl(c, &out->from_v, c.d_from_v, NO));
    W2_TRY(dl(c, &out->from_e, c.d_from_e, NO));
    W2_TRY(dl(c, &out->to_off, c.d_to_off, NV + 1));
    W2_TRY(dl(c, &out->to_v, c.d_to_v, NO));
    W2_TRY(dl(c, &out->to_e, c.d_to_e, NO));
    W2_TRY(dl(c, &out->fwd_xlat, c.d_fwdX, E));
    W2_TRY(dl(c, &out->rev_xlat, c.d_revX, E));
    if (c.pathed_done) {
        out->n_paths = c.n;
        W2_TRY(dl(c, &out->path_offset, c.d_path_offset, c.n));
        W2_TRY(dl(c, &out->path_off, c.d_path_off, c.n + 1));
        W2_TRY(dl(c, &out->path_edges, c.d_path_edges, c.path_total));
    }
    W2_HIP(hipStreamSynchronize(st));
    if (!NV) { out->from_off[0] = 0; out->to_off[0] = 0; }
    c.release(d_len); c.release(d_nb); c.release(d_boff); c.release(d_packed);
    return 0;
}

int w2rap_step2_shard_begin(w2rap_step2_ctx* h, uint32_t rank, uint32_t world, const uint64_t* solid_per_rank, uint32_t n_buckets, uint32_t n_passes, uint64_t M, uint64_t D,
                            const uint64_t* hist101, const w2rap_edge_hint* hint) {
    if (!h || !solid_per_rank) return W2RAP_E_ARG;
    Ctx& c = h->c;
    W2_HIP(hipSetDevice(c.device));
    if (hint) {
        if (hint->n_edges && (!hint->packed || !hint->byte_off || !hint->len)) { c.err = "edge_order_hint: null array"; return W2RAP_E_HINT; }
        for (uint64_t e = 0; e < hint->n_edges; ++e)
            if (hint->byte_off[e + 1] < hint->byte_off[e] || hint->byte_off[e + 1] - hint->byte_off[e] != ((uint64_t)hint->len[e] + 3) / 4) {
                c.err = "edge_order_hint: byte_off does not match len"; return W2RAP_E_HINT;
            }
    }
    c.M = M; c.D = D;
    if (hist101) for (int i = 0; i < 101; ++i) c.hist[i] = hist101[i];
    return shard_begin(c, rank, world, solid_per_rank, n_buckets, n_passes ? n_passes : 1, hint);
}
int w2rap_step2_local_dict_slice(w2rap_step2_ctx* h, uint64_t n_solid, uint64_t expected_total) {
    if (!h) return W2RAP_E_ARG;
    Ctx& c = h->c;
    W2_HIP(hipSetDevice(c.device));
    return local_dict_slice(c, n_solid, expected_total);
}
int w2rap_step2_shard_next(w2rap_step2_ctx* h, w2rap_xchg* x) {
    if (!h || !x) return W2RAP_E_ARG;
    Ctx& c = h->c;
    W2_HIP(hipSetDevice(c.device));
    Timer t(c.stream);
    const int rc = shard_next(c, x);
    c.ms_graph += t.stop();
    c.presolve();
    return rc;
}
int w2rap_step2_shard_recv(w2rap_step2_ctx* h, const uint64_t* recv_count, uint32_t elem_bytes, void** d_recv) {
    if (!h || !recv_count || !d_recv || !elem_bytes) return W2RAP_E_ARG;
    (void)hipSetDevice(h->c.device);
    return shard_recv(h->c, recv_count, elem_bytes, d_recv);
}
int w2rap_step2_shard_host_words(w2rap_step2_ctx* h, const uint64_t* words) { return h && words ? shard_host_words(h->c, words) : W2RAP_E_ARG; }
int w2rap_step2_shard_info(w2rap_step2_ctx* h, uint64_t out[8]) { return h && out ? shard_info(h->c, out) : W2RAP_E_ARG; }
uint64_t w2rap_step2_device_bytes(w2rap_step2_ctx* h) {
    if (!h) return 0;
    uint64_t b = 0;
    for (auto& x : h->c.sizes) b += x.second;
    return b;
}
uint64_t w2rap_step2_device_peak_bytes(w2rap_step2_ctx* h, int reset) {
    if (!h) return 0;
    const uint64_t p = h->c.peak_bytes;
    if (reset) h->c.peak_bytes = h->c.live_bytes;
    return p;
}

void w2rap_step2_free(w2rap_step2_out* o) {
    if (!o) return;
    void* ps[] = {o->edge_packed, o->edge_byte_off, o->edge_len, o->vleft, o->vright, o->from_off, o->from_v, o->from_e,
                  o->to_off, o->to_v, o->to_e, o->fwd_xlat, o->rev_xlat, o->path_offset, o->path_off, o->path_edges};
    for (void* p : ps) std::free(p);
    std::memset(o, 0, sizeof(*o));
}

}  // extern "C"
