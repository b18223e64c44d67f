// segment_quads.hpp -- host side of k_xcorr_segments_quad: cover a window's station pairs with "quads"
// (templates a, b) x (signals c, d).  One quad costs two forward transforms per segment whatever the number of its
// four products (a,c), (a,d), (b,c), (b,d) that are wanted, so the cover should use as few quads as possible:
// greedy, always the quad that takes the most pairs still uncovered (4 when one exists; a leftover pair is joined with
// any other leftover pair -- two wanted products, the cost of the pair-at-a-time kernel).  All pairs of S stations
// need about P/2 quads for large S (S = 3: 1 quad for 3 pairs, 8: 8 for 28, 16: 32 for 120).
// The orientation of a pair (template first) is the caller's and is never flipped: it fixes the sign of the lag.
#pragma once

#include <array>
#include <utility>
#include <vector>

namespace tdoa {

constexpr int kMaxQuadStations = 32;     // the greedy search is O(P S^2) per quad; 32 stations (496 pairs) take milliseconds

struct StationQuad {
    int a, b, c, d;        // station numbers; b = -1 / d = -1: slot empty
    int pair[4];           // index into the caller's pair list of (a,c), (a,d), (b,c), (b,d); -1 = not wanted
};

inline std::vector<StationQuad> build_segment_quads(int n_stations, const std::vector<std::pair<int, int>> &pairs)
{
    const int S = n_stations;
    std::vector<int> rem((size_t)S * S, -1);     // rem[t * S + s] = index of the uncovered pair (t, s)
    int left = 0;
    for (size_t i = 0; i < pairs.size(); i++) {
        const int a = pairs[i].first, c = pairs[i].second;
        if (a < 0 || c < 0 || a >= S || c >= S || rem[(size_t)a * S + c] >= 0) continue;    // caller validates; duplicates once
        rem[(size_t)a * S + c] = (int)i;
        left++;
    }
    auto has = [&](int t, int s) { return t >= 0 && s >= 0 && rem[(size_t)t * S + s] >= 0; };
    std::vector<StationQuad> out;
    while (left > 0) {
        int best = 0, qa = -1, qb = -1, qc = -1, qd = -1;
        for (int a = 0; a < S && best < 4; a++)
            for (int c = 0; c < S && best < 4; c++) {
                if (!has(a, c)) continue;
                if (best == 0) { best = 1; qa = a; qc = c; qb = qd = -1; }
                for (int b = 0; b < S && best < 4; b++) {
                    if (b == a) continue;
                    const int bc = has(b, c) ? 1 : 0;
                    for (int d = 0; d < S; d++) {
                        if (d == c) continue;
                        const int n = 1 + (has(a, d) ? 1 : 0) + bc + (has(b, d) ? 1 : 0);
                        if (n > best) { best = n; qa = a; qb = b; qc = c; qd = d; if (n == 4) break; }
                    }
                }
            }
        StationQuad q{qa, qb, qc, qd, {-1, -1, -1, -1}};
        const int ts[4] = {qa, qa, qb, qb}, ss[4] = {qc, qd, qc, qd};
        for (int o = 0; o < 4; o++)
            if (has(ts[o], ss[o])) {
                q.pair[o] = rem[(size_t)ts[o] * S + ss[o]];
                rem[(size_t)ts[o] * S + ss[o]] = -1;
                left--;
            }
        // a slot whose products are all unwanted stays empty (no loads, no arithmetic on its frames)
        if (q.pair[2] < 0 && q.pair[3] < 0) q.b = -1;
        if (q.pair[1] < 0 && q.pair[3] < 0) q.d = -1;
        out.push_back(q);
    }
    return out;
}

}  // namespace tdoa
