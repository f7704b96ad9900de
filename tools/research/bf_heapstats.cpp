// bf_heapstats.cpp -- research: replays the lower-queue trace of bf_trace.cpp on libstdc++'s heap and measures the structure
// a level-synchronous exact replay could exploit.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <vector>
struct Rec { uint32_t op, prio, x, y; };
struct Ent { int prio; uint32_t x, y; uint64_t id; };
struct Cmp { bool operator()(const Ent& a, const Ent& b) const { return a.prio > b.prio; } };
int main(int argc, char** argv)
{
    FILE* f = fopen(argv[1], "rb");
    int scan_lo = argc > 2 ? atoi(argv[2]) : 6, scan_hi = argc > 3 ? atoi(argv[3]) : 25;
    std::vector<Rec> all; Rec r;
    while (fread(&r, sizeof r, 1, f) == 1) all.push_back(r);
    uint64_t levels = 0, lv_vevent = 0, pops = 0, vevents = 0, climbs = 0, pushes = 0, order_mismatch_levels = 0, nonmono = 0;
    uint64_t sift_total = 0, sift_cnt = 0, tdepth_total = 0, dup_levels = 0, stale_pops = 0;
    std::map<int, uint64_t> mhist, nhist;
    uint64_t maxn = 0;
    size_t i = 0; int scan = -1;
    std::vector<Ent> h;   // lower heap
    uint64_t next_id = 1;
    bool in_update = false; int cur_level = -1;
    // level bookkeeping
    std::vector<uint64_t> predicted; size_t pred_at = 0; bool level_has_v = false, level_bad = false;
    auto start_level = [&](int d) {
        // T_d and its right-first preorder
        predicted.clear(); pred_at = 0; level_has_v = false; level_bad = false;
        std::vector<size_t> st; st.push_back(0);
        while (!st.empty()) {
            size_t q = st.back(); st.pop_back();
            predicted.push_back(h[q].id);
            size_t l = 2 * q + 1, rr = 2 * q + 2;
            if (l < h.size() && h[l].prio == d) st.push_back(l);      // left pushed first -> right popped first
            if (rr < h.size() && h[rr].prio == d) st.push_back(rr);
        }
        int m = (int)predicted.size();
        mhist[m < 8 ? m : (m < 16 ? 8 : (m < 32 ? 16 : (m < 64 ? 32 : (m < 128 ? 64 : (m < 256 ? 128 : 256)))))]++;
        // duplicates of a cell among the level's entries
        std::vector<uint64_t> cells;
        std::vector<size_t> st2; st2.push_back(0);
        while (!st2.empty()) { size_t q = st2.back(); st2.pop_back(); cells.push_back(((uint64_t)h[q].x << 32) | h[q].y);
            size_t l = 2 * q + 1, rr = 2 * q + 2; if (l < h.size() && h[l].prio == d) st2.push_back(l); if (rr < h.size() && h[rr].prio == d) st2.push_back(rr); }
        std::sort(cells.begin(), cells.end());
        if (std::adjacent_find(cells.begin(), cells.end()) != cells.end()) ++dup_levels;
        ++levels;
    };
    auto end_level = [&]() {
        if (cur_level < 0) return;
        if (level_has_v) ++lv_vevent;
        if (level_bad && !level_has_v) ++order_mismatch_levels;
        cur_level = -1;
    };
    for (; i < all.size(); ++i) {
        const Rec& e = all[i];
        if (e.op == 100) { scan = (int)e.prio; continue; }
        const bool count = scan >= scan_lo && scan <= scan_hi;
        if (e.op == 0) { in_update = true; cur_level = -1; continue; }
        if (e.op == 6) { if (count) end_level(); cur_level = -1; in_update = false; if (!h.empty()) { fprintf(stderr, "heap not empty at end\n"); return 1; } continue; }
        if (e.op == 1) {   // push lower
            Ent x{(int)e.prio, e.x, e.y, next_id++};
            h.push_back(x);
            // climb?
            size_t pos = h.size() - 1;
            bool cl = pos > 0 && h[(pos - 1) / 2].prio > x.prio;
            std::push_heap(h.begin(), h.end(), Cmp());
            if (count && in_update && cur_level >= 0) { ++pushes; if (cl) ++climbs; }
            if (count) maxn = std::max<uint64_t>(maxn, h.size());
            continue;
        }
        if (e.op == 4 || e.op == 5) {
            const int d = (int)e.prio;
            if (count) {
                if (d != cur_level) { if (d < cur_level) ++nonmono; end_level(); cur_level = d; start_level(d); nhist[(int)(h.size() / 256)]++; }
                ++pops; if (e.op == 4) ++stale_pops;
                // v-event?
                if (h.size() > 1 && h.back().prio == d) { ++vevents; level_has_v = true; }
                if (pred_at >= predicted.size() || predicted[pred_at] != h[0].id) level_bad = true;
                ++pred_at;
                // sift length below T_d: simulate
                {
                    size_t len = h.size() - 1; const Ent v = h.back();
                    size_t hole = 0; int td = 0, moved = 0;
                    while (true) {
                        size_t c = 2 * hole + 2;
                        if (c < len) { if (h[c].prio > h[c - 1].prio) --c; }
                        else if (c == len) { c = len - 1; }
                        else break;
                        if (h[c].prio > v.prio) break;
                        if (h[c].prio == d) ++td; else ++moved;
                        hole = c;
                    }
                    sift_total += moved; ++sift_cnt; tdepth_total += td;
                }
            }
            if (h[0].x != e.x || h[0].y != e.y || h[0].prio != (int)e.prio) { fprintf(stderr, "replay mismatch at %zu\n", i); return 1; }
            std::pop_heap(h.begin(), h.end(), Cmp()); h.pop_back();
            continue;
        }
    }
    printf("scans %d..%d: levels %lu pops %lu (stale %lu) pushes-in-level %lu climbs %lu (%.2f%%)\n", scan_lo, scan_hi, levels, pops, stale_pops, pushes, climbs, 100.0 * climbs / (pushes ? pushes : 1));
    printf("v-events %lu in %lu levels (%.1f%% of levels); order mismatch w/o v-event: %lu levels; non-monotone level switches %lu; levels with duplicate cells %lu\n",
           vevents, lv_vevent, 100.0 * lv_vevent / levels, order_mismatch_levels, nonmono, dup_levels);
    printf("mean pops/level %.1f; mean T_d depth walked per pop %.2f; mean entries moved below T_d per pop %.2f; max heap %lu\n",
           (double)pops / levels, (double)tdepth_total / sift_cnt, (double)sift_total / sift_cnt, maxn);
    printf("m histogram (bucket lower bound: levels):"); for (auto& kv : mhist) printf(" %d:%lu", kv.first, kv.second); printf("\n");
    printf("heap size/256 at level start:"); for (auto& kv : nhist) printf(" %d:%lu", kv.first, kv.second); printf("\n");
    return 0;
}
