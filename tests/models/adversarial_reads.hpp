// adversarial_reads.hpp — structured reads for the host models of the seeded pass (seed_bounds.cpp, seed_band.cpp,
// seed_warmup.cpp): cases the random generators of those models practically never draw.
//
// spacer_case: the read's columns BETWEEN two sampled k-mers hold residues without potential (N under new_dna_matrix(.., b"N"):
// Wp = 0), the reference holds (1) a "far" copy of the read from which the last column of k-mer j, the columns between and the
// first column of k-mer j + 1 are missing — a path along it inserts those columns, and the N columns among them lose gap_extend
// each, not maxw + gap_extend — and (2) an "anchor" copy whose own loss (a deletion of g reference bases between two k-mers) is
// steered to lie around twice the lambda the seeded pass charges for the two broken k-mers. If the pass charges the far path more
// than it really loses, the anchor copy is accepted with a score below the truth. Schemes with maxw - mismatch >= gap_open make
// lambda = gap_open, the case where the insertion run is the cheapest way through a k-mer.
#pragma once
#include <cstdint>
#include <random>
#include <vector>

#include "../../zoe_amd/csrc/zsw_seed.hpp"

namespace adversarial {

// Fills ref and q (residue indices; 4 = the residue without potential). Returns false if L is too short for two k-mers.
template <class Rng>
bool spacer_case(Rng& rng, const zsw::SeedParams& p, int L, std::vector<uint8_t>* ref, std::vector<uint8_t>* q) {
    auto rnd = [&](int lo, int hi) { return lo + (int)(rng() % (uint64_t)(hi - lo + 1)); };
    int m, stride, c0;
    zsw::seed_layout(L, p.K, p.spacer, &m, &stride, &c0);
    if (m < 2) return false;
    q->resize(L);
    for (auto& x : *q) x = (uint8_t)rnd(0, 3);
    // which gaps between k-mers hold N: all of them, or only the one the far copy breaks
    const int j = rnd(0, m - 2);
    const bool all_gaps = rnd(0, 1) == 0;
    for (int i = 0; i + 1 < m; ++i) {
        if (!all_gaps && i != j) continue;
        for (int c = c0 + i * stride + p.K; c < c0 + (i + 1) * stride; ++c) (*q)[c] = 4;
    }
    // far copy: the read without columns [c_j + K - 1, c_{j+1}] (sometimes one column fewer on either side)
    const int cut_lo = c0 + j * stride + p.K - 1 + rnd(0, 1), cut_hi = c0 + (j + 1) * stride + 1 - rnd(0, 1);
    std::vector<uint8_t> far;
    for (int c = 0; c < L; ++c)
        if (c < cut_lo || c >= cut_hi) far.push_back((*q)[c] == 4 ? (uint8_t)rnd(0, 3) : (*q)[c]);
    // anchor copy: the whole read, N columns spelled at random, with g extra reference bases between two k-mers so that the copy
    // loses gap_open + (g - 1) * gap_extend: around 2 * lambda
    const int lam = zsw::seed_lambda(p, stride);
    std::vector<uint8_t> anchor;
    const int target = 2 * lam + rnd(-3, 2);
    int g = p.ge > 0 ? (target - p.go) / p.ge + 1 : 1;
    if (g < 0) g = 0;
    if (g > 40) g = 40;
    const int jg = rnd(0, m - 2);
    const int at = c0 + jg * stride + p.K + (stride - p.K) / 2;  // between k-mer jg and jg + 1
    for (int c = 0; c < L; ++c) {
        if (c == at)
            for (int x = 0; x < g; ++x) anchor.push_back((uint8_t)rnd(0, 3));
        anchor.push_back((*q)[c] == 4 ? (uint8_t)rnd(0, 3) : (*q)[c]);
    }
    if (rnd(0, 3) == 0 && L > 8) anchor[rnd(0, (int)anchor.size() - 1)] = (uint8_t)rnd(0, 3);  // sometimes a substitution on top
    auto junk = [&](int n) {
        for (int x = 0; x < n; ++x) ref->push_back((uint8_t)rnd(0, 3));
    };
    ref->clear();
    junk(rnd(0, 30));
    const bool far_first = rnd(0, 1) == 0;
    const std::vector<uint8_t>&first = far_first ? far : anchor, &second = far_first ? anchor : far;
    ref->insert(ref->end(), first.begin(), first.end());
    junk(rnd(20, 60));
    ref->insert(ref->end(), second.begin(), second.end());
    junk(rnd(0, 30));
    return true;
}

}  // namespace adversarial
