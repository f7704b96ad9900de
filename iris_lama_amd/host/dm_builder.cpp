// dm_builder.cpp -- see dm_builder.hpp: the first build of Loc2D's distance map (addObstacle x N on an empty map, one update()).
#include "dm_builder.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <queue>

namespace lama {
namespace detail {

namespace {

constexpr uint64_t UNIVERSAL_CONSTANT = 2642244ull;      // Map::m2p, /root/reference/include/lama/sdm/map.h:153-161
constexpr uint8_t F_VALID = 1, F_QUEUED = 2, F_ON = 4;   // valid_obstacle, is_queued, Container mask bit

struct Cell {                                            // distance_t (10 B in the reference) without the z offset, plus the mask bit
    int16_t ox, oy;
    uint16_t sq;
    uint8_t flags;
    uint8_t pad;
};
static_assert(sizeof(Cell) == 8, "dense store: 8 bytes per cell");

// queue_pair_t / compare_prio of /root/reference/include/lama/sdm/dynamic_distance_map.h:90-98: ordered by the priority alone
struct Entry { int32_t prio; uint32_t x, y; };
struct ComparePrio { bool operator()(const Entry& l, const Entry& r) const { return l.prio > r.prio; } };

struct Store {
    uint32_t x0 = 0, y0 = 0;          // map coordinates of the store's corner (patch aligned)
    uint32_t w = 0, h = 0;            // cells
    std::vector<Cell> c;
    bool inside(uint32_t x, uint32_t y) const { return x - x0 < w && y - y0 < h; }
    // the non-const Map::get: the patch exists afterwards and the cell's mask bit is set
    Cell& get(uint32_t x, uint32_t y)
    {
        Cell& k = c[(size_t)(y - y0) * w + (x - x0)];
        k.flags |= F_ON;
        return k;
    }
};

}  // namespace

bool build_distance_map(const uint32_t* cells_xy, size_t n, uint32_t max_sqdist, sdm::HostMap& out, uint32_t& processed,
                        uint64_t max_store_cells)
{
    processed = 0;
    if (n == 0 || max_sqdist > 65025u) return false;      // (255 cells: what the reference's uint16_t sqdist and the wide device library hold)
    uint32_t xlo = UINT32_MAX, xhi = 0, ylo = UINT32_MAX, yhi = 0;
    for (size_t i = 0; i < n; ++i) {
        xlo = std::min(xlo, cells_xy[2 * i]); xhi = std::max(xhi, cells_xy[2 * i]);
        ylo = std::min(ylo, cells_xy[2 * i + 1]); yhi = std::max(yhi, cells_xy[2 * i + 1]);
    }
    // what the build can touch: a cell is overwritten while its squared distance is below max_sqdist, and lower() looks at the
    // neighbours of such a cell -- ceil(sqrt(max_sqdist)) + 1 cells beyond an obstacle; one more for good measure, then whole patches
    const uint32_t reach = (uint32_t)std::ceil(std::sqrt((double)max_sqdist)) + 2u;
    if (xlo < reach + 32u || ylo < reach + 32u || xhi > UINT32_MAX - reach - 32u || yhi > UINT32_MAX - reach - 32u) return false;
    Store st;
    st.x0 = ((xlo - reach) >> 5) << 5; st.y0 = ((ylo - reach) >> 5) << 5;
    const uint32_t x1 = (((xhi + reach) >> 5) + 1u) << 5, y1 = (((yhi + reach) >> 5) + 1u) << 5;
    st.w = x1 - st.x0; st.h = y1 - st.y0;
    if ((uint64_t)st.w * st.h > max_store_cells) return false;
    st.c.assign((size_t)st.w * st.h, Cell{0, 0, 0, 0, 0});

    std::priority_queue<Entry, std::vector<Entry>, ComparePrio> lower;
    // addObstacle (:212-226), in the caller's order
    for (size_t i = 0; i < n; ++i) {
        const uint32_t x = cells_xy[2 * i], y = cells_xy[2 * i + 1];
        Cell& cell = st.get(x, y);
        if ((cell.flags & F_VALID) && cell.sq == 0) continue;          // already an obstacle
        cell.sq = 0; cell.ox = 0; cell.oy = 0;
        cell.flags |= F_VALID | F_QUEUED;
        lower.push(Entry{0, x, y});
    }
    // update (:160-197): the raise queue is empty (nothing was removed from an empty map); the lower loop
    static const int DX[4] = {1, 0, -1, 0}, DY[4] = {0, 1, 0, -1};     // deltas_ (:40-43)
    while (!lower.empty()) {
        const Entry e = lower.top();
        lower.pop();
        ++processed;
        Cell& cur = st.get(e.x, e.y);
        if (!(cur.flags & F_VALID)) continue;
        const int64_t obx = (int64_t)e.x + cur.ox, oby = (int64_t)e.y + cur.oy;     // absolute position of the cell's obstacle
        if (!st.inside((uint32_t)obx, (uint32_t)oby)) return false;
        if (st.get((uint32_t)obx, (uint32_t)oby).sq != 0) continue;                 // (:187-188; that cell is no longer an obstacle: cannot happen here)
        // lower (:281-330)
        if (!(cur.flags & F_QUEUED)) continue;          // already processed through a closer obstacle
        for (int i = 0; i < 4; ++i) {
            if (DX[i] * (int)cur.ox > 0 || DY[i] * (int)cur.oy > 0) continue;       // only away from the obstacle
            const int64_t nx = (int64_t)e.x + DX[i], ny = (int64_t)e.y + DY[i];
            if (!st.inside((uint32_t)nx, (uint32_t)ny)) return false;               // (the margin makes this unreachable)
            Cell& nb = st.get((uint32_t)nx, (uint32_t)ny);
            const int64_t ddx = nx - obx, ddy = ny - oby;
            const uint32_t new_sq = (uint32_t)(ddx * ddx + ddy * ddy);
            const uint32_t cmp_sq = (nb.flags & F_VALID) ? nb.sq : max_sqdist;
            bool overwrite = new_sq < cmp_sq;
            if (!overwrite && new_sq == nb.sq) {
                const int64_t qx = nx + nb.ox, qy = ny + nb.oy;                     // the neighbour's own obstacle
                if (!st.inside((uint32_t)qx, (uint32_t)qy)) return false;
                const Cell& ob = st.get((uint32_t)qx, (uint32_t)qy);
                if (!(nb.flags & F_VALID) || !((ob.flags & F_VALID) && ob.sq == 0)) overwrite = true;
            }
            if (overwrite) {
                lower.push(Entry{(int32_t)new_sq, (uint32_t)nx, (uint32_t)ny});
                nb.sq = (uint16_t)new_sq;
                nb.ox = (int16_t)(obx - nx); nb.oy = (int16_t)(oby - ny);
                nb.flags |= F_VALID | F_QUEUED;
            }
        }
        cur.flags &= (uint8_t)~F_QUEUED;
    }

    // the reference's records: every patch a get() touched (Container::alloc zeroes a new patch; a touched cell has its mask bit)
    out.ids.clear(); out.cells.clear(); out.masks.clear();
    const uint32_t pw = st.w >> 5, ph = st.h >> 5;
    for (uint32_t px = 0; px < pw; ++px)
        for (uint32_t py = 0; py < ph; ++py) {
            uint64_t mask[16];
            std::memset(mask, 0, sizeof(mask));
            bool any = false;
            for (uint32_t cy = 0; cy < 32; ++cy) {
                const Cell* row = &st.c[(size_t)(py * 32 + cy) * st.w + px * 32];
                for (uint32_t cx = 0; cx < 32; ++cx)
                    if (row[cx].flags & F_ON) { const uint32_t idx = cx | (cy << 5); mask[idx >> 6] |= 1ull << (idx & 63); any = true; }
            }
            if (!any) continue;
            out.ids.push_back((uint64_t)((st.x0 >> 5) + px) * UNIVERSAL_CONSTANT + ((st.y0 >> 5) + py));
            const size_t base = out.cells.size();
            out.cells.resize(base + 10240, 0);
            for (uint32_t cy = 0; cy < 32; ++cy) {
                const Cell* row = &st.c[(size_t)(py * 32 + cy) * st.w + px * 32];
                for (uint32_t cx = 0; cx < 32; ++cx) {
                    const Cell& k = row[cx];
                    if (!(k.flags & F_ON)) continue;
                    uint8_t* o = &out.cells[base + 10 * (size_t)(cx | (cy << 5))];
                    std::memcpy(o, &k.ox, 2); std::memcpy(o + 2, &k.oy, 2);     // obstacle[0], [1]; [2] stays 0
                    std::memcpy(o + 6, &k.sq, 2);
                    o[8] = (k.flags & F_VALID) ? 1 : 0; o[9] = (k.flags & F_QUEUED) ? 1 : 0;
                }
            }
            out.masks.insert(out.masks.end(), mask, mask + 16);
        }
    return true;
}

}  // namespace detail
}  // namespace lama
