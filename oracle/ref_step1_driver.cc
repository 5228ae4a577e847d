// oracle/ref_step1_driver.cc -- TEST INFRASTRUCTURE, not product code.
//
// Runs the UNMODIFIED reference's Step 1 exactly as its main() does (src/modules/w2rap-contigger.cc:305-316):
//   ExtractReads(read_files, out_dir, subsam_names, subsam_starts, &bases, &quals);
//   bases.WriteAll(out_dir + "/frag_reads_orig.fastb"); quals.WriteAll(out_dir + "/frag_reads_orig.qualp");
// Used to make the Step-1 goldens (tests/golden/make_golden.py step1) and as the CPU baseline of bench.py --step1.
//
// usage: ref_step1 <out_dir> <r1.fastq,r2.fastq> [threads=1]
#include <omp.h>
#include <cstdlib>
#include <iostream>
#include <string>
#include "Basevector.h"
#include "feudal/PQVec.h"
#include "paths/long/large/ExtractReads.h"
#include "system/System.h"

int main(int argc, char** argv) {
    if (argc < 3) { std::cerr << "usage: ref_step1 out_dir r1.fastq,r2.fastq [threads]\n"; return 2; }
    std::string out_dir = argv[1], reads = argv[2];
    int threads = argc > 3 ? atoi(argv[3]) : 1;
    omp_set_num_threads(threads);
    vecbvec bases; VecPQVec quals;
    vec<String> subsam_names; vec<int64_t> subsam_starts;
    double t0 = WallClockTime();
    ExtractReads(String(reads), String(out_dir), subsam_names, subsam_starts, &bases, &quals);
    double t1 = WallClockTime();
    bases.WriteAll(out_dir + "/frag_reads_orig.fastb");
    quals.WriteAll(out_dir + "/frag_reads_orig.qualp");
    std::cout << "REF_STEP1 reads " << bases.size() << " extract_s " << (t1 - t0) << " write_s " << (WallClockTime() - t1) << std::endl;
    return 0;
}
