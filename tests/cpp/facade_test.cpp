// Compiled and run by tests/test_cpp_facade.py.  Mirrors the doc example of src/lib.rs:57-68 and the KAT loop
// of tests/main.rs:60-73: prints one line per k-min-mer so the Python side can diff against the oracle.
#include "s2k.hpp"

#include <cstdio>
#include <fstream>
#include <iostream>
#include <string>

int main(int argc, char **argv) {
    if (argc < 6) {
        std::fprintf(stderr, "usage: %s <seq-file> <l> <k> <density> <mode 0..3>\n", argv[0]);
        return 2;
    }
    std::ifstream f(argv[1], std::ios::binary);
    std::string seq((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    try {
        s2k::Engine eng(0);
        s2k::KminmersIterator it(eng, seq, std::stoul(argv[2]), std::stoul(argv[3]), std::stod(argv[4]),
                                 (s2k::HashMode)std::stoi(argv[5]));
        for (s2k::KminmerHash km : it)
            std::printf("%llu %zu %zu %zu %d\n", (unsigned long long)km.get_hash(), km.start, km.end, km.offset, (int)km.rev);
        if (argc > 6) { // minimizer-triple iterators over the same sequence: "M <mode> start end hash" lines
            const uint32_t bound = s2k::hash_bound(std::stod(argv[4]));
            for (s2k::Minimizer m : s2k::NtHashHPCIterator(eng, seq, std::stoul(argv[2]), bound)) std::printf("M 1 %zu %zu %u\n", m.start, m.end, m.hash);
            for (s2k::Minimizer m : s2k::NtHashSIMDIterator(eng, seq, std::stoul(argv[2]), bound)) std::printf("M 2 %zu %zu %u\n", m.start, m.end, m.hash);
            for (s2k::Minimizer m : s2k::NtHashHPCSIMDIterator(eng, seq, std::stoul(argv[2]), bound)) std::printf("M 3 %zu %zu %u\n", m.start, m.end, m.hash);
        }
        { // two engines linked for double buffering (s2k_chain_after through the facade): linking, self-linking refused, unlinking
            s2k::Engine other(0);
            eng.chain_after(&other);
            other.chain_after(&eng);
            try {
                eng.chain_after(&eng);
                return 5;
            } catch (const s2k::Error &e) {
                if (e.status != S2K_ERR_INVALID_ARG) return 6;
            }
            s2k::KminmersIterator again(other, seq, std::stoul(argv[2]), std::stoul(argv[3]), std::stod(argv[4]), (s2k::HashMode)std::stoi(argv[5]));
            size_t n_again = 0, n_first = 0;
            for (s2k::KminmerHash km : again) n_again += 1 + 0 * km.start;
            for (s2k::KminmerHash km : it) n_first += 1 + 0 * km.start;
            if (n_again != n_first) return 7;
            eng.chain_after(nullptr);
            other.chain_after(nullptr);
        }
        // error behaviour: k == 0 panics in the reference (src/lib.rs:246), throws here
        try {
            s2k::KminmersIterator bad(eng, seq, 31, 0, 0.01, s2k::HashMode::Regular);
            return 3;
        } catch (const s2k::Error &e) {
            if (e.status != S2K_ERR_K_RANGE) return 4;
        }
    } catch (const s2k::Error &e) {
        std::fprintf(stderr, "s2k error %d: %s\n", (int)e.status, e.what());
        return 1;
    }
    return 0;
}
