// zsw_driver — C++ host driver over include/zoe_sw.hpp: reads FASTQ records and a one-sequence FASTA/plain reference,
// aligns every read on the GPU (sw_align_from_i8, w256 preset: what `Nucleotides::into_local_profile` users call) and prints
// SAM records with the fields the reference's SamData::from_alignment fills (src/data/records/sam/mod.rs:223-245):
// POS = ref_range.start + 1, CIGAR = states, AS:i = score.
//
//   g++ -O2 -std=c++17 -Iinclude examples/zsw_driver.cpp -o examples/zsw_driver -Lzoe_amd -lzoe_sw_hip -Wl,-rpath,$PWD/zoe_amd
//   ./examples/zsw_driver reference.txt reads.fastq [--score-only | --3pass]
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

#include "zoe_sw.hpp"

static std::string read_reference(const std::string& path, std::string* name) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error("cannot open " + path);
    std::string line, seq;
    *name = "ref";
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (!line.empty() && line[0] == '>') {
            std::istringstream ss(line.substr(1));
            ss >> *name;
        } else {
            seq += line;
        }
    }
    return seq;
}

int main(int argc, char** argv) {
    if (argc < 3) {
        std::cerr << "usage: zsw_driver reference.(fa|txt) reads.fastq [--score-only | --3pass]\n";
        return 2;
    }
    const bool score_only = argc > 3 && !std::strcmp(argv[3], "--score-only");
    const bool three_pass = argc > 3 && !std::strcmp(argv[3], "--3pass");  // sw_align_from_i8_3pass instead of sw_align_from_i8
    try {
        std::string ref_name;
        const std::string reference = read_reference(argv[1], &ref_name);
        std::ifstream fq(argv[2]);
        if (!fq) throw std::runtime_error(std::string("cannot open ") + argv[2]);
        std::vector<std::string> names, reads, quals;
        std::string h, s, p, q;
        while (std::getline(fq, h) && std::getline(fq, s) && std::getline(fq, p) && std::getline(fq, q)) {
            if (h.empty() || h[0] != '@') throw std::runtime_error("malformed FASTQ header: " + h);
            names.push_back(h.substr(1, h.find_first_of(" \t") == std::string::npos ? std::string::npos : h.find_first_of(" \t") - 1));
            reads.push_back(s);
            quals.push_back(q);
        }
        zoe::GpuContext ctx(0);
        const zoe::WeightMatrix weights = zoe::WeightMatrix::new_dna_matrix(2, -5, 'N');
        zoe::LocalProfilesBatch profiles(ctx, reads, weights, -10, -1);
        if (score_only) {
            auto scores = profiles.sw_score_from_i8(reference);
            for (size_t i = 0; i < reads.size(); ++i)
                std::cout << names[i] << '\t' << (scores[i].is_some() ? std::to_string(scores[i].value) : std::string("*")) << '\n';
            return 0;
        }
        auto alns = three_pass ? profiles.sw_align_from_i8_3pass(reference) : profiles.sw_align_from_i8(reference);
        std::cout << "@HD\tVN:1.6\n@SQ\tSN:" << ref_name << "\tLN:" << reference.size() << '\n';
        for (size_t i = 0; i < reads.size(); ++i) {
            if (alns[i].is_some()) {
                const zoe::Alignment& a = alns[i].value;
                std::cout << names[i] << "\t0\t" << ref_name << '\t' << a.ref_start + 1 << "\t255\t" << a.cigar() << "\t*\t0\t0\t" << reads[i]
                          << '\t' << quals[i] << "\tAS:i:" << a.score << '\n';
            } else {
                std::cout << names[i] << "\t4\t*\t0\t0\t*\t*\t0\t0\t" << reads[i] << '\t' << quals[i] << '\n';
            }
        }
    } catch (const std::exception& e) {
        std::cerr << "zsw_driver: " << e.what() << '\n';
        return 1;
    }
    return 0;
}
