// oracle/ref_driver.cc -- TEST INFRASTRUCTURE, not product code.
//
// A ~60-line driver (ours) that links against the UNMODIFIED reference
// translation units where they lie under /root/reference/src and runs the
// reference's own Step 2 exactly as its main() does
// (src/modules/w2rap-contigger.cc:326-346):
//   bases.ReadAll(fastb); quals.ReadAll(qualp);
//   buildReadQGraph(bases,quals,false,false,minQual,minFreq,.75,0,&hbv,&paths,60,out_dir,"",0);
//   FixPaths(hbv,paths);
//   BinaryWriter::writeFile(<out>/<prefix>.small_K.hbv, hbv);
//   WriteReadPathVec(paths, <out>/<prefix>.small_K.paths);
// No reference source is copied; this file only calls the reference's public
// functions.  Built by oracle/Makefile into oracle/_ref/ref_step2 (gitignored).
//
// usage: ref_step2 <out_dir> <prefix> [threads=1] [min_qual=7] [min_freq=4]
//   reads <out_dir>/frag_reads_orig.{fastb,qualp}; writes <out_dir>/<prefix>.small_K.{hbv,paths},
//   <out_dir>/small_K.freqs, and prints "REF_TIME buildReadQGraph <s> FixPaths <s>".
#include <omp.h>
#include <sys/time.h>
#include <cstdlib>
#include <iostream>
#include <string>
#include "Basevector.h"
#include "feudal/PQVec.h"
#include "feudal/BinaryStream.h"
#include "paths/HyperBasevector.h"
#include "paths/long/ReadPath.h"
#include "paths/long/BuildReadQGraph.h"
#include "paths/long/large/GapToyTools.h"

static double now() { timeval t; gettimeofday(&t, 0); return t.tv_sec + 1e-6 * t.tv_usec; }

int main(int argc, char** argv) {
    if (argc < 3) { std::cerr << "usage: ref_step2 out_dir prefix [threads] [min_qual] [min_freq]\n"; return 2; }
    std::string out_dir = argv[1], prefix = argv[2];
    int threads = argc > 3 ? atoi(argv[3]) : 1;
    unsigned minQual = argc > 4 ? atoi(argv[4]) : 7;
    unsigned minFreq = argc > 5 ? atoi(argv[5]) : 4;
    omp_set_num_threads(threads);
    vecbvec bases; VecPQVec quals;
    bases.ReadAll(out_dir + "/frag_reads_orig.fastb");
    quals.ReadAll(out_dir + "/frag_reads_orig.qualp");
    HyperBasevector hbv; ReadPathVec paths;
    double t0 = now();
    buildReadQGraph(bases, quals, false, false, minQual, minFreq, .75, 0, &hbv, &paths, 60, out_dir, "", 0);
    double t1 = now();
    FixPaths(hbv, paths);
    double t2 = now();
    BinaryWriter::writeFile(out_dir + "/" + prefix + ".small_K.hbv", hbv);
    WriteReadPathVec(paths, (out_dir + "/" + prefix + ".small_K.paths").c_str());
    std::cout << "REF_TIME buildReadQGraph " << (t1 - t0) << " FixPaths " << (t2 - t1)
              << " threads " << threads << " reads " << bases.size() << std::endl;
    return 0;
}
