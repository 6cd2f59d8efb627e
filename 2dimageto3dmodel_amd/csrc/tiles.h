// Tile geometry shared by the binning kernel (proj_transform.hip) and the renderers (proj_render*.hip).
// A tile is TH x TW rays (silhouette pixels); a ray gets LPR lanes with D depths each (LPR*D >= S).
#pragma once

namespace m355 {

struct TileShape {
    int lpr, d, th, tw;
};

inline bool tile_shape(int S, TileShape &c)
{
    if (S < 2) return false;
    if (S <= 64) c = {16, 4, 8, 8};
    else if (S <= 128) c = {16, 8, 8, 8};
    else if (S <= 256) c = {32, 8, 4, 8};
    else if (S <= 512) c = {64, 8, 4, 4};
    else return false;
    return true;
}

inline int tile_count(int S)
{
    TileShape c;
    if (!tile_shape(S, c)) return -1;
    return ((S + c.tw - 1) / c.tw) * ((S + c.th - 1) / c.th);
}

}  // namespace m355
