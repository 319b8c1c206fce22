// sage2_amd/csrc/synth.cpp -- repo-owned deterministic synthetic read generator
// (SURVEY.md section 8d): uniform random genome, optional planted repeats, paired-end
// fragments (insert = 3L, mate 2 = reverse complement of the fragment's far end),
// interleaved output, optional substitution errors and mixed read lengths.
// Counter-based (splitmix64 of (seed, stream, index)) so any slice of the read set can be
// produced independently, on any rank, in any order.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "sage2ov.h"

namespace {
inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
inline uint64_t rnd(uint64_t seed, uint64_t stream, uint64_t idx) {
    return splitmix64(splitmix64(seed * 0xD1342543DE82EF95ULL + stream) + idx);
}
enum { S_GENOME = 1, S_FRAG = 2, S_LEN = 3, S_ERR = 4, S_REPPOS = 5, S_REPSRC = 6 };
const char kBase[4] = {'A', 'C', 'G', 'T'};
}  // namespace

extern "C" {

int sage2ov_synth_genome(const sage2ov_synth_params* p, uint8_t* genome) {
    if (!p || !genome || p->genome_len == 0) return SAGE2OV_ERR_ARG;
    const uint64_t G = p->genome_len;
    for (uint64_t q = 0; q < (G + 31) / 32; q++) {
        uint64_t r = rnd(p->seed, S_GENOME, q);
        for (uint64_t b = 0; b < 32 && q * 32 + b < G; b++) genome[q * 32 + b] = (r >> (2 * b)) & 3;
    }
    // planted repeats: repeat family f is a segment of the genome copied to n_copies places
    for (uint32_t f = 0; f < p->n_repeat_families; f++) {
        if (p->repeat_len == 0 || p->repeat_len >= G) break;
        uint64_t src = rnd(p->seed, S_REPSRC, f) % (G - p->repeat_len);
        std::vector<uint8_t> unit(genome + src, genome + src + p->repeat_len);
        for (uint32_t c = 0; c < p->repeat_copies; c++) {
            uint64_t dst = rnd(p->seed, S_REPPOS, (uint64_t)f * 1000003ULL + c) % (G - p->repeat_len);
            memcpy(genome + dst, unit.data(), p->repeat_len);
        }
    }
    return SAGE2OV_OK;
}

uint32_t sage2ov_synth_read_len(const sage2ov_synth_params* p, uint64_t r) {
    if (p->read_len_min == 0 || p->read_len_min >= p->read_len) return p->read_len;
    return p->read_len_min + (uint32_t)(rnd(p->seed, S_LEN, r) % (p->read_len - p->read_len_min + 1));
}

// reads [first, first+n) as ASCII, back to back in `bases`; offsets has n+1 entries.
int sage2ov_synth_reads_ascii(const sage2ov_synth_params* p, const uint8_t* genome, uint64_t first, uint64_t n,
                              char* bases, uint64_t* offsets) {
    if (!p || !genome || !bases || !offsets) return SAGE2OV_ERR_ARG;
    const uint64_t G = p->genome_len, L = p->read_len, insert = 3 * L;
    if (G < insert) return SAGE2OV_ERR_ARG;
    uint64_t pos = 0;
    for (uint64_t x = 0; x < n; x++) {
        const uint64_t r = first + x, pair = r >> 1, mate = r & 1;
        const uint64_t fr = rnd(p->seed, S_FRAG, pair);
        const uint64_t start = fr % (G - insert + 1);
        const bool flip = (fr >> 63) & 1;          // fragment taken from the reverse strand
        const uint32_t len = sage2ov_synth_read_len(p, r);
        offsets[x] = pos;
        // fragment coordinates: f in [0, insert); forward-strand fragment base f = genome[start+f];
        // reverse-strand fragment base f = comp(genome[start+insert-1-f]).
        // mate 1 = fragment[0..len), mate 2 = revcomp(fragment[insert-len..insert)).
        for (uint32_t b = 0; b < len; b++) {
            uint64_t f; bool comp;
            if (mate == 0) { f = b; comp = false; } else { f = insert - 1 - b; comp = true; }
            uint64_t g; if (!flip) g = start + f; else { g = start + insert - 1 - f; comp = !comp; }
            uint8_t base = genome[g]; if (comp) base = 3 - base;
            if (p->err_ppm) {
                uint64_t e = rnd(p->seed, S_ERR, r * 4096 + b);
                if ((e % 1000000ULL) < p->err_ppm) base = (base + 1 + ((e >> 40) % 3)) & 3;
            }
            bases[pos++] = kBase[base];
        }
    }
    offsets[n] = pos;
    return SAGE2OV_OK;
}

// interleaved FASTA (>r<pair>/1, >r<pair>/2), one sequence line per read
int sage2ov_synth_write_fasta(const sage2ov_synth_params* p, const char* path) {
    if (!p || !path) return SAGE2OV_ERR_ARG;
    std::vector<uint8_t> genome(p->genome_len);
    int rc = sage2ov_synth_genome(p, genome.data()); if (rc) return rc;
    FILE* f = fopen(path, "w"); if (!f) return SAGE2OV_ERR_IO;
    const uint64_t chunk = 1 << 16;
    std::vector<char> bases(chunk * p->read_len); std::vector<uint64_t> off(chunk + 1);
    for (uint64_t first = 0; first < p->n_reads; first += chunk) {
        uint64_t n = std::min<uint64_t>(chunk, p->n_reads - first);
        rc = sage2ov_synth_reads_ascii(p, genome.data(), first, n, bases.data(), off.data());
        if (rc) { fclose(f); return rc; }
        for (uint64_t x = 0; x < n; x++) {
            uint64_t r = first + x;
            fprintf(f, ">r%llu/%d\n", (unsigned long long)(r >> 1), (int)(r & 1) + 1);
            fwrite(bases.data() + off[x], 1, off[x + 1] - off[x], f); fputc('\n', f);
        }
    }
    fclose(f);
    return SAGE2OV_OK;
}
}
