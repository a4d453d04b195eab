// s2k_main -- command-line driver, the counterpart of the reference's src/main.rs:
//   no arguments : demo on a test sequence (src/main.rs:13-48): HPC string, then the k-min-mers of the four
//                  HashModes at l=28, k=5, d=0.1
//   <fastx> [l k density mode batch_Mbp] : file mode (src/main.rs:51-83; defaults l=31 k=5 d=0.01 Regular like
//                  :53-60) -- streams the file through the GPU and prints the k-min-mer count and the wall time.
//                  (The reference's second argument, the thread count, has no meaning here.)
#include "../../include/s2k.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

static const char *mode_name(int m) { return m == 0 ? "Regular" : m == 1 ? "Hpc" : m == 2 ? "Simd" : "HpcSimd"; }

int main(int argc, char **argv) {
    try {
        s2k::Engine eng(0);
        if (argc < 2) {
            const std::string seq = "AACTGCACTGCACTGCACTGCACACTGCACTGCACTGCACTGCACACTGCACTGCACTGACTGCACTGCACTGCACTGCACTGCCTGC";
            std::printf("seq:    \"%s\"\n", seq.c_str());
            std::printf("Demonstrating how to construct k-min-mers (l=28, k=5, d=0.1) out of a test sequence\n");
            for (int m : {0, 2, 1, 3}) { // Regular, Simd, Hpc, HpcSimd -- the order of src/main.rs:40
                std::printf("mode: %s\n", mode_name(m));
                s2k::KminmersIterator it(eng, seq, 28, 5, 0.1, (s2k::HashMode)m);
                for (s2k::KminmerHash km : it)
                    std::printf("kminmer: KminmerHash { hash: %llu, start: %zu, end: %zu, offset: %zu, rev: %s }\n",
                                (unsigned long long)km.hash, km.start, km.end, km.offset, km.rev ? "true" : "false");
            }
            return 0;
        }
        s2k_params p{31, 5, 0.01, S2K_MODE_REGULAR, 0};
        if (argc > 2) p.l = (uint32_t)std::strtoul(argv[2], nullptr, 10);
        if (argc > 3) p.k = (uint32_t)std::strtoul(argv[3], nullptr, 10);
        if (argc > 4) p.density = std::strtod(argv[4], nullptr);
        if (argc > 5) p.mode = std::atoi(argv[5]);
        uint64_t batch = argc > 6 ? std::strtoull(argv[6], nullptr, 10) * 1000000ull : 0;
        std::printf("Enumerating k-min-mers for the input file %s on GPU 0 (l=%u k=%u d=%g mode=%s)\n", argv[1], p.l, p.k,
                    p.density, mode_name(p.mode));
        s2k_counts tot;
        double sec = 0;
        s2k_status st = s2k_run_file(eng.raw(), argv[1], &p, batch, &tot, &sec);
        if (st != S2K_OK) {
            std::fprintf(stderr, "error: %s (%s)\n", s2k_strerror(st), s2k_last_error(eng.raw()));
            return 1;
        }
        std::printf("reads: %llu bases: %llu minimizers: %llu kminmers: %llu xor: %llu\n", (unsigned long long)tot.n_reads,
                    (unsigned long long)tot.n_bases, (unsigned long long)tot.n_minimizers, (unsigned long long)tot.n_kminmers,
                    (unsigned long long)tot.xor_hash);
        std::printf("FASTA to kminmers in %.3fs.\n", sec);
    } catch (const s2k::Error &e) {
        std::fprintf(stderr, "s2k error %d: %s\n", (int)e.status, e.what());
        return 1;
    }
    return 0;
}
