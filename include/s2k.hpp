// s2k.hpp -- header-only C++17 facade over the C ABI (s2k.h) with the reference's interface shape:
//   s2k::KminmersIterator(seq, l, k, density, mode)  ~  KminmersIterator::new      src/lib.rs:89-131
//   range-for over it yields s2k::KminmerHash           ~  impl Iterator             src/lib.rs:179-270
//   s2k::KminmerHash{hash,start,end,offset,rev}, == / < by hash only               src/kminmer.rs:128-135,181-204
//   s2k::HashMode                                                                    src/lib.rs:21-27
//   s2k::NtHashHPCIterator / NtHashSIMDIterator / NtHashHPCSIMDIterator (seq, l, hash_bound)   src/lib.rs:6-13 (re-exports)
// Parameter violations that panic in the reference throw s2k::Error here.  The efficient entry point is
// Engine::extract (a batch of reads per call); the per-read iterator exists for reference-style call sites.
#pragma once
#include "s2k.h"

#include <cstdint>
#include <stdexcept>
#include <string>
#include <string_view>
#include <vector>

namespace s2k {

enum class HashMode : int32_t { Regular = S2K_MODE_REGULAR, Hpc = S2K_MODE_HPC, Simd = S2K_MODE_SIMD, HpcSimd = S2K_MODE_HPCSIMD };

struct Error : std::runtime_error {
    s2k_status status;
    Error(s2k_status st, const std::string &what) : std::runtime_error(what), status(st) {}
};

struct KminmerHash { // src/kminmer.rs:128-135
    uint64_t hash;
    size_t start, end, offset;
    bool rev;
    uint64_t get_hash() const { return hash; }                                      // src/kminmer.rs:162-164
    friend bool operator==(const KminmerHash &a, const KminmerHash &b) { return a.hash == b.hash; } // :181-185
    friend bool operator<(const KminmerHash &a, const KminmerHash &b) { return a.hash < b.hash; }    // :194-198
};

class Batch { // owns one s2k_result
  public:
    Batch() { res_ = s2k_result{}; }
    Batch(const Batch &) = delete;
    Batch &operator=(const Batch &) = delete;
    Batch(Batch &&o) noexcept : res_(o.res_) { o.res_ = s2k_result{}; }
    Batch &operator=(Batch &&o) noexcept {
        if (this != &o) {
            s2k_result_free(&res_);
            res_ = o.res_;
            o.res_ = s2k_result{};
        }
        return *this;
    }
    ~Batch() { s2k_result_free(&res_); }
    uint64_t n_reads() const { return res_.n_reads; }
    uint64_t n_kminmers() const { return res_.n_kminmers; }
    uint64_t begin_of(uint64_t r) const { return res_.km_off[r]; }
    uint64_t end_of(uint64_t r) const { return res_.km_off[r + 1]; }
    KminmerHash item(uint64_t r, uint64_t i) const { // i-th k-min-mer of read r
        uint64_t g = res_.km_off[r] + i;
        return KminmerHash{res_.hash[g], res_.start[g], res_.end[g], (size_t)i, res_.rev[g] != 0};
    }
    uint64_t n_minimizers() const { return res_.n_minimizers; } // with S2K_FLAG_WANT_MINIMIZERS
    const s2k_result &raw() const { return res_; }
    s2k_result *out() { return &res_; }

  private:
    s2k_result res_;
};

class Engine { // one per thread and device (like one KminmersIterator per thread, src/main.rs:65-79)
  public:
    explicit Engine(int device = 0) {
        s2k_status st;
        ctx_ = s2k_create(device, &st);
        if (!ctx_) throw Error(st, std::string("s2k_create: ") + s2k_strerror(st));
    }
    Engine(const Engine &) = delete;
    Engine &operator=(const Engine &) = delete;
    ~Engine() { s2k_destroy(ctx_); }

    // double buffering with two engines on one device (s2k_chain_after): this engine's minimizer kernels wait for those of prev's most recent
    // call; chain both ways, alternate the calls from one thread; nullptr unlinks (do so before prev is destroyed)
    void chain_after(Engine *prev) {
        s2k_status st = s2k_chain_after(ctx_, prev ? prev->ctx_ : nullptr);
        if (st != S2K_OK) throw Error(st, s2k_strerror(st));
    }
    // bases per sub-batch of extract(): H2D / kernels / D2H of consecutive sub-batches overlap (0 = default, 2^29)
    void set_host_batch(uint64_t bases) {
        s2k_status st = s2k_set_host_batch(ctx_, bases);
        if (st != S2K_OK) throw Error(st, s2k_strerror(st));
    }
    Batch extract(const uint8_t *bases, const uint64_t *read_off, uint64_t n_reads, size_t l, size_t k, double density,
                  HashMode mode, uint32_t flags = 0) {
        s2k_params p{(uint32_t)l, (uint32_t)k, density, (int32_t)mode, flags};
        Batch b;
        s2k_status st = s2k_extract(ctx_, bases, read_off, n_reads, &p, b.out());
        if (st != S2K_OK) throw Error(st, std::string(s2k_strerror(st)) + ": " + s2k_last_error(ctx_));
        return b;
    }
    Batch extract(const std::vector<std::string_view> &reads, size_t l, size_t k, double density, HashMode mode, uint32_t flags = 0) {
        std::vector<uint64_t> off(reads.size() + 1, 0);
        std::string bases;
        for (size_t i = 0; i < reads.size(); i++) {
            bases.append(reads[i]);
            off[i + 1] = bases.size();
        }
        return extract(reinterpret_cast<const uint8_t *>(bases.data()), off.data(), reads.size(), l, k, density, mode, flags);
    }
    s2k_ctx *raw() { return ctx_; }

  private:
    s2k_ctx *ctx_;
};

class KminmersIterator { // src/lib.rs:70-131: per-read facade
  public:
    KminmersIterator(Engine &eng, std::string_view seq, size_t l, size_t k, double density, HashMode mode)
        : batch_(eng.extract(std::vector<std::string_view>{seq}, l, k, density, mode)) {}
    struct iterator {
        const Batch *b;
        uint64_t i;
        KminmerHash operator*() const { return b->item(0, i); }
        iterator &operator++() {
            ++i;
            return *this;
        }
        bool operator!=(const iterator &o) const { return i != o.i; }
    };
    iterator begin() const { return iterator{&batch_, 0}; }
    iterator end() const { return iterator{&batch_, batch_.n_kminmers()}; }
    size_t size() const { return (size_t)batch_.n_kminmers(); }

  private:
    Batch batch_;
};

// The crate's minimizer iterators (re-exported at src/lib.rs:6-13), over S2K_FLAG_WANT_MINIMIZERS.  Like the reference they take
// the u32 hash bound, not a density.  One GPU call per sequence: for throughput use Engine::extract with the flag on a batch.
// l > seq.size() throws as KSizeOutOfRange does (src/nthash_hpc.rs:117-121); seq.size() == l is accepted as in the reference.
struct Minimizer { // NtHashHPCIterator::Item = (start, end, hash), src/nthash_hpc.rs:193
    size_t start, end;
    uint32_t hash;
};
template <HashMode MODE>
class MinimizerIterator {
  public:
    MinimizerIterator(Engine &eng, std::string_view seq, size_t l, uint32_t hash_bound) {
        if (l > seq.size()) throw Error(S2K_ERR_L_RANGE, "K size is out of range for the given sequence size"); // src/nthash_hpc.rs:19-22
        if (seq.size() == l) {
            // Accepted by the reference's iterators (only l > len is KSizeOutOfRange).  The scalar Hpc iterator yields nothing then (its only
            // l-mer is the last one, src/nthash_hpc.rs:265-267); the SIMD ones may yield the single l-mer.  The kernels follow
            // KminmersIterator (len <= l: nothing, src/lib.rs:97), so that l-mer is computed on the GPU on the sequence plus one base
            // that starts a new run, and only what lies inside the original sequence is kept (end_).
            if (MODE == HashMode::Hpc) return; // empty batch
            std::string ext(seq);
            ext.push_back(seq.back() == 'A' ? 'C' : 'A');
            batch_ = eng.extract(std::vector<std::string_view>{ext}, l, 1, s2k_density_for_bound(hash_bound), MODE, S2K_FLAG_WANT_MINIMIZERS);
            const s2k_result &r = batch_.raw();
            end_ = (r.n_minimizers >= 1 && r.mn_j[0] == 0 && r.mn_jend[0] < l) ? 1 : 0;
            limited_ = true;
            return;
        }
        batch_ = eng.extract(std::vector<std::string_view>{seq}, l, 1, s2k_density_for_bound(hash_bound), MODE, S2K_FLAG_WANT_MINIMIZERS);
    }
    struct iterator {
        const s2k_result *r;
        uint64_t i;
        Minimizer operator*() const { return Minimizer{r->mn_j[i], r->mn_jend[i], r->mn_hash[i]}; }
        iterator &operator++() {
            ++i;
            return *this;
        }
        bool operator!=(const iterator &o) const { return i != o.i; }
    };
    iterator begin() const { return iterator{&batch_.raw(), 0}; }
    iterator end() const { return iterator{&batch_.raw(), size()}; }
    size_t size() const { return limited_ ? (size_t)end_ : (size_t)batch_.n_minimizers(); }

  private:
    Batch batch_;
    uint64_t end_ = 0;     // seq.size() == l: how many of the batch's minimizers lie inside the original sequence (0 or 1)
    bool limited_ = false;
};
using NtHashHPCIterator = MinimizerIterator<HashMode::Hpc>;         // src/nthash_hpc.rs:99-283: `<=`, last HPC l-mer dropped, end = last base of the last run
using NtHashHPCSIMDIterator = MinimizerIterator<HashMode::HpcSimd>; // src/nthash_hpc_simd.rs:17-68: `<` vs f32 bound, end = start of the last run
using NtHashSIMDIterator = MinimizerIterator<HashMode::Simd>;       // src/nthash_avx512_32.rs:14-164: Item = (pos, hash): use .start and .hash

inline uint32_t hash_bound(double density) { return s2k_hash_bound(density); } // src/lib.rs:91

} // namespace s2k
