// Host check of gpsig_amd/csrc/sig_pieces.hpp: the depth pieces of the feature contraction cover the slabs exactly once, in order, the
// equal ones follow the round-3 formula, the graded tail halves.  Prints the number of violations.
#include <cstdio>
#include <vector>

#include "sig_pieces.hpp"

int main() {
    int bad = 0;
    std::vector<int> b(300);
    for (int nslab : {1, 2, 7, 8, 16, 31, 64, 100, 213, 1000, 2341, 20000})
        for (int equal : {1, 2, 3, 4, 11, 16, 31, 128})
            for (int graded : {1, 2, 3, 4, 5}) {
                const int n = gpsig::sig_piece_bounds(nslab, equal, graded, b.data());
                const int eq = equal > nslab ? nslab : equal;
                if (b[0] != 0 || b[n] != nslab) ++bad;
                for (int s = 0; s < n; ++s)
                    if (b[s + 1] < b[s] || (nslab >= eq && b[s + 1] == b[s] && graded == 1)) ++bad;
                for (int s = 0; s < eq; ++s)
                    if (b[s] != int((long long)nslab * s / eq)) ++bad;                       // the equal pieces: round 3's boundaries
                const int len = nslab - b[eq - 1];
                const bool cut = graded > 1 && len >= 2 * graded;
                if (n != eq - 1 + (cut ? graded : 1)) ++bad;
                if (cut) {
                    for (int s = eq - 1; s + 2 < n; ++s) {                                   // each graded piece: half of what was left
                        const int a = b[s + 1] - b[s], rest = nslab - b[s];
                        if (a != rest / 2) ++bad;
                    }
                    if (b[n] - b[n - 1] < 1) ++bad;
                }
            }
    // BASELINE configs[1]: 2341 slabs, 11 equal pieces, the last one as 1/2, 1/4, 1/8, 1/8
    const int n = gpsig::sig_piece_bounds(2341, 11, 4, b.data());
    if (n != 14 || b[10] != 2128 || b[11] != 2128 + 106 || b[12] != 2128 + 106 + 53 || b[13] != 2128 + 106 + 53 + 27 || b[14] != 2341) ++bad;
    printf("%d\n", bad);
    return 0;
}
