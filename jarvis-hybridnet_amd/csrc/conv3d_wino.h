// Shared declarations of the Winograd 3D convolution kernels (conv3d_wino.hip: one role per
// workgroup, one tile each; conv3d_wino_pw.hip: persistent, wave-specialised).
#pragma once
#include <vector>
#include "jh_common.h"

namespace jh {



constexpr int kWTY = 8, kWTX = 8;                           // (y, x) outputs per workgroup
constexpr int kWPY = kWTY + 2, kWPX = kWTX + 2;

// ---- tile shapes for volumes whose (y, x) extent is not a multiple of 8 (round 5) ----------------------------------
// The 16 MFMA rows of a z-slice are 16 Winograd tiles of 2 x 2 outputs: a 4 x 4 block (8 x 8 voxels) everywhere a whole
// block fits -- and, where an extent leaves a remainder of 1..4 voxels (the reference's shipped 72^3 grid runs V2V at 36^3
// and 18^3, projects/Example_Project/config.yaml:36-37: 36 = 4 x 8 + 4, 18 = 2 x 8 + 2), the remainder strip is tiled with
// blocks that are 2 tiles wide: 8 x 2 tiles = 16 x 4 voxels down the right edge, 2 x 8 tiles = 4 x 16 voxels along the
// bottom edge (it owns the corner).  36^2: 16 + 2 + 3 = 21 blocks per slice instead of 25, 18^2: 4 + 1 + 2 = 7 instead of
// 9.  A remainder of 5..7 keeps the 4 x 4 block (at least 5/8 full).  Every shape aligns its tiles to even coordinates,
// so a voxel's 4 x 4 input patch -- and with it every output bit -- does not depend on the shape that computes it.
// `lc` = log2 of the tiles per block row: 2 (4 x 4), 1 (8 x 2), 3 (2 x 8).
struct WinoTiling {
  int n44x, n44y;          // 4 x 4-tile blocks: a n44y x n44x grid from the origin
  int nR, nB;              // blocks of the right strip (x0 = W44, y0 = 16 i) and of the bottom strip (y0 = H44, x0 = 16 i)
  int H44, W44;            // where the strips start = the extent the 4 x 4 blocks own (<= H, W)
  int slabs;               // z-slabs of tz slices
  int shaped;              // any strip at all
};
inline WinoTiling wino_tiling(int D, int H, int W, int tz) {
  WinoTiling g;
  const int rh = H % 8, rw = W % 8;
  const bool sy = rh >= 1 && rh <= 4 && H > 8, sx = rw >= 1 && rw <= 4 && W > 8;
  g.H44 = sy ? H - rh : H;
  g.W44 = sx ? W - rw : W;
  g.n44y = (g.H44 + 7) / 8;
  g.n44x = (g.W44 + 7) / 8;
  g.nR = sx ? (g.H44 + 15) / 16 : 0;
  g.nB = sy ? (W + 15) / 16 : 0;
  g.slabs = (D + tz - 1) / tz;
  g.shaped = sx || sy;
  // JH_WINO_SHAPES: 0 = never (the plain grid of 4 x 4 blocks), 2 = always (measurement: the shaped kernels on a volume
  // that has no strips)
  const int knob = JH_ENV_KNOB("JH_WINO_SHAPES");
  if (knob == 0) g.shaped = 0;
  if (knob == 2) g.shaped = 1;
  return g;
}
inline int wino_blocks_per_slab(const WinoTiling& g) { return g.n44x * g.n44y + g.nR + g.nB; }

struct WinoArgs {
  const float* x;          // [N][D][H][W][cin_p]
  float* y;                // [N][D][H][W][cout_p] raw output
  const float* u;          // transformed weights [16 f][3 dz][cin_p/8][cout_p16/16][64][2]
  const float* bias;       // [cout_p16] or nullptr
  const double* in_stats;  // InstanceNorm (+ in_act) of the input applied on load, or nullptr
  float in_inv;
  int in_act;
  double* stats;           // [N][cout_p][2] or nullptr
  int N, D, H, W, cin_p, cout_p, cout_p16;
  int abl;                 // experiment knob (JH_WS_ABL), 0 in production
  long long* dbg;          // per-phase cycle sums of workgroup 0 (JH_WINO_DBG), nullptr in production
  WinoTiling tiling;       // block shapes of a z-slice (filled by launch_conv3d_wino)
  const int* shape_tab;    // tiling.shaped: the loader's per-shape patch tables, [3][kWinoShapeWords] (device)
};

struct WinoTile { int n, z0, y0, x0, lc, hlim, wlim; };    // hlim / wlim: outputs of this block at y >= hlim or x >= wlim
                                                            // belong to another block (or lie outside the volume)
// Tile r of the N * slabs * blocks-per-slab tiles of one column-block group, SHAPE-major (all 4 x 4 blocks of all
// images first, then the right strips, then the bottom strips: a workgroup that walks the list with a fixed stride
// changes shape at most twice), then image, z-slab, block.
#if defined(__HIPCC__)
__host__ __device__
#endif
inline WinoTile wino_decode(const WinoTiling& g, int N, int H, int W, int tz, int r) {
  const int n44 = g.n44x * g.n44y;
  const int A = N * g.slabs * n44, B = N * g.slabs * g.nR;
  WinoTile t;
  int cnt;
  if (r < A) { t.lc = 2; cnt = n44; }
  else if (r < A + B) { r -= A; t.lc = 1; cnt = g.nR; }
  else { r -= A + B; t.lc = 3; cnt = g.nB; }
  const int idx = r % cnt;
  r /= cnt;
  t.z0 = (r % g.slabs) * tz;
  t.n = r / g.slabs;
  if (t.lc == 2) {
    t.x0 = (idx % g.n44x) * 8; t.y0 = (idx / g.n44x) * 8;
    t.hlim = g.H44; t.wlim = g.W44;
  } else if (t.lc == 1) {
    t.x0 = g.W44; t.y0 = idx * 16;
    t.hlim = g.H44; t.wlim = W;
  } else {
    t.y0 = g.H44; t.x0 = idx * 16;
    t.hlim = H; t.wlim = W;
  }
  return t;
}
// Both kernels decode their tiles arithmetically (a per-tile table LOAD cost an lgkmcnt(0) wait at every tile boundary of
// every wave: 0.65 vs 0.57 ms).  The persistent kernel's loader wave takes its per-lane patch tables, one set per block
// shape (index lc - 1), from a table: 21 rounds
// of 64 float4 items cover the 6 x 108 x 2 items of the largest patch; words [k][lane], k < 21: byte offset of item
// (lane + 64 k) relative to the patch origin, k = 21 .. 27: the lane's item masks z0, z1, y0, y1, x0, x1 (first / beyond-
// last patch plane of the volume's first / last block along that axis) and `tail` (items beyond the patch).
constexpr int kWinoShapeIter = 21, kWinoShapeWords = (kWinoShapeIter + 7) * 64;
// host: [3 shapes][kWinoShapeWords] ints for a D x H x W volume with cin_p input channels (empty when the volume has no
// remainder strips)
std::vector<int> wino_tables(int D, int H, int W, int cin_p);

// persistent form; returns -1 when the launch should fall back to the one-role kernel
int launch_conv3d_wino_pw(const WinoArgs& a, int nr, hipStream_t s);

}  // namespace jh
