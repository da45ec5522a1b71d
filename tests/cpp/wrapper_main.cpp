// Drives thermite::ThermiteAligner (include/thermite.hpp) the way cellranger drives the
// reference's wrapper (src/wrapper.rs:64-101): one read per call, SAM records out.
//   wrapper_main <index file> <min_seed_len> <min_aln_score> <fastq>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>

#include "thermite.hpp"

int main(int argc, char** argv) {
  if (argc != 5) return 2;
  try {
    thermite::ThermiteAligner a(argv[1]);
    a.opts_mut().min_seed_len = (std::size_t)atoi(argv[2]);
    a.opts_mut().min_aln_score = atoi(argv[3]);
    fputs(a.header_view().c_str(), stdout);
    std::ifstream f(argv[4]);
    std::string name, seq, plus, qual;
    std::size_t n_stripped = 0, n_tagged = 0;
    while (std::getline(f, name) && std::getline(f, seq) && std::getline(f, plus) && std::getline(f, qual)) {
      for (const auto& rec : a.align_read_with_tags(name.substr(1), seq, qual)) puts(rec.c_str());
      // the cellranger-facing call: the same records without TX / GX / GN / RE (src/wrapper.rs:136-139)
      for (const auto& rec : a.align_read(name.substr(1), seq, qual)) {
        n_stripped++;
        if (rec.find("\tTX:Z:") != std::string::npos || rec.find("\tGX:Z:") != std::string::npos ||
            rec.find("\tGN:Z:") != std::string::npos || rec.find("\tRE:A:") != std::string::npos)
          n_tagged++;
        const bool unmapped = rec.find("\t4\t*\t0\t") != std::string::npos;  // unmapped_sam_record carries no tags at all
        if (!unmapped && (rec.find("\tAS:i:") == std::string::npos || rec.find("\tnM:i:") == std::string::npos)) n_tagged += 1000;
      }
    }
    fprintf(stderr, "stripped_records %zu still_tagged %zu\n", n_stripped, n_tagged);
    fprintf(stderr, "est_mem %zu\n", thermite::ThermiteAligner::est_mem(argv[1]));
  } catch (const thermite::Error& e) {
    fprintf(stderr, "error %d: %s\n", e.code, e.what());
    return 1;
  }
  return 0;
}
