// BuildReadQGraph_gpu.cc -- the translation unit a w2rap-contigger maintainer compiles INSTEAD of src/paths/long/BuildReadQGraph.cc
// (and links with -lw2rap_step2): the reference's own signature (src/paths/long/BuildReadQGraph.h:24-29), called by its unmodified
// main at src/modules/w2rap-contigger.cc:338, bound to the C ABI of include/w2rap_step2.h.
//
// It flattens the feudal containers (vecbvec -> .fastb packing, src/feudal/FieldVec.h:768; VecPQVec -> raw qualities through the
// reference's own PQVec::unpack, src/feudal/PQVec.h:86-91 -- PQVec keeps its bytes private), makes ONE call, and refills the caller's
// objects through the reference's public methods: AddEdge in object-id order reproduces the adjacency ordering rule
// (src/graph/DigraphTemplate.h:1829-1839).  Errors are fatal in the reference's style (FatalErr); the library never throws or exits.
// The GPUs to use: environment W2RAP_GPUS (default 1) -- the reference's call has no such argument (-t sets its OpenMP threads).
//
// oracle/Makefile target `gpu_contigger` builds oracle/_ref/w2rap-contigger-gpu = the reference's main + all its objects with this
// file in BuildReadQGraph.o's place; tests/test_gpu_shim.py runs it.
#include <cstdlib>
#include <string>
#include <vector>

#include "paths/long/BuildReadQGraph.h"
#include "paths/HyperBasevector.h"
#include "paths/long/ReadPath.h"
#include "feudal/PQVec.h"
#include "Basevector.h"
#include "Qualvector.h"
#include "system/System.h"
#include "w2rap_step2.h"

void buildReadQGraph(vecbvec const& reads, VecPQVec const& quals, bool /*doFillGaps*/, bool /*doJoinOverlaps*/,
                     unsigned minQual, unsigned minFreq, double /*minFreq2Fract*/, unsigned /*maxGapSize*/,
                     HyperBasevector* pHBV, ReadPathVec* pPaths, int _K, std::string workdir, std::string /*tmpdir*/,
                     unsigned char /*disk_batches*/) {
    const size_t n = reads.size();
    // 1. flatten: offsets first, then the bytes in parallel (the reference's OpenMP pool is idle here)
    std::vector<uint64_t> boff(n + 1, 0), qoff(n + 1, 0);
    std::vector<uint32_t> len(n);
    for (size_t i = 0; i < n; ++i) {
        len[i] = reads[i].size();
        boff[i + 1] = boff[i] + (len[i] + 3) / 4;
        qoff[i + 1] = qoff[i] + len[i];
    }
    std::vector<uint8_t> bases(boff[n] + 1, 0), q(qoff[n] + 1, 0);
    #pragma omp parallel
    {
        qvec qv;
        #pragma omp for schedule(dynamic, 4096)
        for (size_t i = 0; i < n; ++i) {
            bvec const& b = reads[i];
            uint8_t* dst = &bases[boff[i]];
            for (unsigned j = 0; j < len[i]; ++j) dst[j >> 2] |= (uint8_t)(b[j] << (2 * (j & 3)));
            quals[i].unpack(&qv);
            for (unsigned j = 0; j < len[i] && j < qv.size(); ++j) q[qoff[i] + j] = qv[j];
        }
    }
    w2rap_reads R{};
    R.n_reads = n; R.bases_packed = bases.data(); R.base_byte_off = boff.data(); R.read_len = len.data();
    R.quals = q.data(); R.qual_off = qoff.data(); R.mem = W2RAP_MEM_HOST;
    const std::string freqs = workdir + "/small_K.freqs";
    w2rap_step2_params P{};
    P.K = (uint32_t)_K; P.min_qual = minQual; P.min_freq = minFreq; P.device = 0;
    P.freqs_path = workdir.empty() ? nullptr : freqs.c_str();
    if (const char* g = std::getenv("W2RAP_GPUS")) P.n_gpus = std::atoi(g);
    if (const char* g = std::getenv("W2RAP_PASSES")) P.n_passes = std::atoi(g);      // hash-range passes of the counting phase (the reference's --disk_batches)
    if (!pPaths) P.flags |= W2RAP_F_GRAPH_ONLY;                                      // the caller wants the graph alone (BuildReadQGraph.cc:1300-1307): no read is pathed
    w2rap_step2_out O{};
    char err[1024] = {0};
    if (w2rap_step2_run(&R, &P, &O, err, sizeof err)) FatalErr("w2rap_step2_run: " << err);
    std::cout << Date() << ": " << O.n_kmers_solid << " / " << O.n_kmers_distinct << " kmers with Freq >= " << minFreq << std::endl;
    // 2. *pHBV exactly as buildHBVFromEdges fills it (src/paths/long/HBVFromEdges.cc:129-151)
    pHBV->Clear(); pHBV->SetK(_K); pHBV->AddVertices(O.n_vertices);
    for (uint64_t e = 0; e < O.n_edge_objs; ++e) {
        bvec s(O.edge_len[e]);
        const uint8_t* p = O.edge_packed + O.edge_byte_off[e];
        for (uint32_t j = 0; j < O.edge_len[e]; ++j) s.set(j, (p[j >> 2] >> (2 * (j & 3))) & 3);
        pHBV->AddEdge(O.vleft[e], O.vright[e], s);
    }
    // 3. *pPaths (FixPaths has been applied: the caller's own FixPaths, w2rap-contigger.cc:340, finds nothing to cut)
    if (pPaths) {
        // (one spare, default-constructed element behind the paths: the reference's FragDist reads paths[id1 + 1] without a bound,
        // src/paths/long/large/GapToyTools3.cc:624-627 -- with an odd number of reads that is paths[size()], which its own
        // buildReadQGraph happens to leave as zeroed heap; an empty path there is what keeps that read of theirs harmless)
        pPaths->clear(); pPaths->reserve(O.n_paths + 2); pPaths->resize(O.n_paths + 1); pPaths->resize(O.n_paths);
        for (uint64_t r = 0; r < O.n_paths; ++r) {
            ReadPath& rp = (*pPaths)[r];
            rp.setOffset(O.path_offset[r]);
            rp.assign(O.path_edges + O.path_off[r], O.path_edges + O.path_off[r + 1]);
        }
        std::cout << Date() << ": " << O.n_reads_pathed << " / " << O.n_paths << " reads pathed, " << O.n_reads_multipathed
                  << " spanning junctions" << std::endl;
    }
    w2rap_step2_free(&O);
}
