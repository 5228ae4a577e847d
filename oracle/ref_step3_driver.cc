// oracle/ref_step3_driver.cc -- TEST INFRASTRUCTURE, not product code.
//
// Runs the UNMODIFIED reference's Step 3 on a Step-2 result, exactly as its main() does after
// `--from_step 3` (src/modules/w2rap-contigger.cc:352-378):
//   BinaryReader::readFile(<out>/<prefix>.small_K.hbv, &hbv); LoadReadPathVec(paths, ....small_K.paths);
//   vecbvec edges(hbv.Edges()); hbv.Involution(inv); FragDist(...);
//   RepathInMemory(hbv, edges, inv, paths, hbv.K(), large_K, hbvr, pathsr, True, True, extend_paths=false);
//   writes <out>/<prefix>.large_K.{hbv,paths}
// Used by the tests to show that OUR small_K graph + paths are a drop-in for the reference's own
// consumer: Step 3 must produce the same large-K graph from either (modulo edge numbering).
//
// usage: ref_step3 <out_dir> <prefix> [large_K=200] [threads=1] [extend_paths=0]      (extend_paths: --extend_paths, w2rap-contigger.cc:371)
#include <omp.h>
#include <cstdlib>
#include <iostream>
#include <string>
#include "Basevector.h"
#include "feudal/BinaryStream.h"
#include "paths/HyperBasevector.h"
#include "paths/long/ReadPath.h"
#include "paths/long/large/GapToyTools.h"
#include "paths/long/large/Repath.h"

int main(int argc, char** argv) {
    if (argc < 3) { std::cerr << "usage: ref_step3 out_dir prefix [large_K] [threads] [extend_paths]\n"; return 2; }
    std::string out_dir = argv[1], prefix = argv[2];
    int large_K = argc > 3 ? atoi(argv[3]) : 200;
    int threads = argc > 4 ? atoi(argv[4]) : 1;
    const bool extend_paths = argc > 5 && atoi(argv[5]) != 0;
    omp_set_num_threads(threads);
    HyperBasevector hbv, hbvr; ReadPathVec paths, pathsr; vec<int> inv;
    BinaryReader::readFile(out_dir + "/" + prefix + ".small_K.hbv", &hbv);
    LoadReadPathVec(paths, (out_dir + "/" + prefix + ".small_K.paths").c_str());
    vecbvec edges(hbv.Edges().begin(), hbv.Edges().end());
    hbv.Involution(inv);
    FragDist(hbv, inv, paths, out_dir + "/" + prefix + ".first.frags.dist");
    pathsr.resize(paths.size());
    RepathInMemory(hbv, edges, inv, paths, hbv.K(), large_K, hbvr, pathsr, True, True, extend_paths);
    BinaryWriter::writeFile(out_dir + "/" + prefix + ".large_K.hbv", hbvr);
    WriteReadPathVec(pathsr, (out_dir + "/" + prefix + ".large_K.paths").c_str());
    std::cout << "REF_STEP3 edges " << hbvr.EdgeObjectCount() << " vertices " << hbvr.N() << " paths " << pathsr.size() << std::endl;
    return 0;
}
