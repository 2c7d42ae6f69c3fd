// seq_pk2_kernel instantiations (float32, two y sequences packed per pair group) and their lookup.
#include "seq_pk2_kernel.hpp"

namespace gpsig {
typedef hipError_t (*SeqLaunchFn)(const SeqGramArgs&, int, size_t, hipStream_t);

// (G, C, D, num_levels): a pair group of G lanes x C columns holds y records of up to G * C rows
#define GPSIG_PK2_SHAPES(X) X(16, 4, 8, 4) X(16, 4, 8, 5) X(64, 2, 16, 5) X(64, 2, 16, 6) X(64, 2, 8, 5)

struct SeqPk2Shape { int G, C, D, M; };
static const SeqPk2Shape PK2_TABLE[] = {
#define X_ROW(G_, C_, D_, M_) {G_, C_, D_, M_},
    GPSIG_PK2_SHAPES(X_ROW)
#undef X_ROW
};

// cheapest built shape for y records of `rows` rows, d features, exactly M levels; false if none
bool seq_pk2_select(int rows, int d, int M, int* G, int* C, int* D) {
    long best = -1;
    for (const SeqPk2Shape& s : PK2_TABLE) {
        if (s.M != M || s.D < d || s.G * s.C < rows) continue;
        const long cost = long(s.G) * s.C * s.D * 2 + (s.G == 16 ? 0 : 1);
        if (best < 0 || cost < best) { best = cost; *G = s.G; *C = s.C; *D = s.D; }
    }
    return best >= 0;
}

// pack: 2 = two y sequences per pair group (f2), 1 = one (float);  waves: wavefronts per workgroup on one x ring (1 or 4)
SeqLaunchFn seq_pk2_lookup(int G, int C, int D, int M, int mode, int pack, int waves) {
#define X_PICK(V_, W_, G_, C_, D_, M_) \
    return mode == MODE_INC ? &seq_pk2_launch<V_, W_, G_, C_, D_, M_, MODE_INC> : &seq_pk2_launch<V_, W_, G_, C_, D_, M_, MODE_PT_DIFF>;
#define X_CASE(G_, C_, D_, M_)                                            \
    if (G == G_ && C == C_ && D == D_ && M == M_) {                       \
        if (pack == 2 && waves == 1) { X_PICK(f2, 1, G_, C_, D_, M_) }    \
        if (pack == 2 && waves == 4) { X_PICK(f2, 4, G_, C_, D_, M_) }    \
        if (pack == 1 && waves == 1) { X_PICK(float, 1, G_, C_, D_, M_) } \
        if (pack == 1 && waves == 4) { X_PICK(float, 4, G_, C_, D_, M_) } \
    }
    GPSIG_PK2_SHAPES(X_CASE)
#undef X_CASE
#undef X_PICK
    return nullptr;
}
}  // namespace gpsig
