// Persistent, wave-specialised Winograd 3D convolution (same mathematics, packed weights and
// output as conv3d_wino.hip; V2V Res3DBlocks, jarvis/hybridnet/v2vnet.py:27-43).
//
// One workgroup per CU (8 waves, two per SIMD, fixed roles) loops over a static list of
// 4 x 8 x 8-voxel output tiles:
//
//   waves 0-3 "matrix waves" : the MFMA stream and nothing else.  Wave w owns the frequencies
//       fy = w, fx = 0..3 of all 4 z-slices and NR column blocks (48 accumulator tiles for
//       NR = 3).  Operand registers are software-pipelined in program order (sched_barrier):
//       the weights of the next frequency step are requested (buffer loads, scalar offsets)
//       as soon as the 24 MFMAs of a z tap have been issued, the LDS operands of the next step
//       go to a second register set.  At the end of a tile the wave folds fx (A^T along x) and
//       DUMPS the partial sums to LDS in two halves -- that is all it sees of the epilogue.
//   waves 4-7 "staging waves": everything else, one channel pass AHEAD of the matrix waves and
//       across tile boundaries: global loads of the raw 6 x 10 x 10 x 8-channel patch,
//       InstanceNorm(+act) on load (mean / rstd from the producer's fused statistics), commit to
//       LDS, the input transform B^T d B into the double-buffered operand array V -- and the
//       whole epilogue of the PREVIOUS tile (A^T along y across the four matrix waves' dumps,
//       bias, statistics, 4 x 4 quad transposes, 16-byte stores, fp64 atomics) while the matrix
//       waves already run the next tile's MFMAs.
//
// So prologue, input transforms and epilogue -- 45 % of a workgroup's life in the one-role
// kernel -- run under MFMAs; the matrix cores wait only for the two dump hand-overs per tile.
//
// Barriers per tile (all 8 waves; `wg_barrier` waits for LDS traffic only, so global loads in
// flight stay in flight): one per channel pass, four for the dump hand-over.  With g = global
// pass index (tile k, pass p):
//   matrix    : [MFMAs of pass g from V[g & 1] | Be] x P   dump(ox 0) Bd1 . Bd2 dump(ox 1) Bd3 . Bd4
//   loader    : [commit pass g+2 -> R[g & 1], request pass g+3          | Be] x P
//   transform : [R[(g+1) & 1] -> V[(g+1) & 1]                           | Be] x P
//   all staging waves: in pass 0 the ox = 1 half of the previous tile's epilogue, then its
//               statistics atomics;  at the tile end Bd1 read(ox 0) Bd2 finish(ox 0) Bd3 read(ox 1) Bd4
// Measured (JH_WINO_DBG cycle counters, 46 -> 46 @ 32^3): while a matrix wave streams fp32 MFMAs
// the staging wave of the same SIMD issues about one instruction per MFMA slot (45-50 cycles),
// whatever its type or priority -- so the staging instruction COUNT per pass is budgeted against
// the 288 MFMAs of a pass: buffer loads with scalar offsets and out-of-range zero padding (no
// address or mask arithmetic), packed fp32 math, mode dispatch outside the item loops.
// LDS: R [2][608][8] 39 KB, V [2][6][16][16][8] 98 KB (XOR-swizzled channel quads: an A
// operand's 16 tile rows x 4 k-quads hit 64 distinct banks), statistics scratch, mean / rstd
// table: 139 KB; the dump uses the V buffer of the pass just consumed (49 KB for NR = 3).
#include <cstdlib>
#include <type_traits>
#include "conv_mfma.h"
#include "conv3d_wino.h"

#ifndef JH_PW_PRIO
#define JH_PW_PRIO 3
#endif
namespace jh {

namespace {
__device__ __forceinline__ void wg_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
constexpr int kPTZ = 4, kPPZ = kPTZ + 2;
constexpr int kPVSZ = kPPZ * 16 * 16 * 8;                   // floats per V buffer
// patch pixels per workgroup tile: 6 z-slices of 10 x 10 (the plain grid of 4 x 4-tile blocks) or of up to 18 x 6
// (SH: the block shapes of conv3d_wino.h), and the floats of an R buffer (19 / 21 rounds of 64 float4 items)
constexpr int pw_pnp(bool sh) { return kPPZ * (sh ? 108 : kWPY * kWPX); }
constexpr int pw_prs(bool sh) { return (pw_pnp(sh) + 8 + (sh ? 16 : 0)) * 8; }

struct Tile { int n, nb0, z0, y0, x0, lc, hlim, wlim; };
typedef float f32x2 __attribute__((ext_vector_type(2)));
// debug timeline: T(slot) adds the cycles since the previous stamp to slot `slot`
#define JH_T(slot) do { if (DBG && dbg_on) { const long long _t = __builtin_readcyclecounter(); dbg_acc[slot] += _t - dbg_t; dbg_t = _t; } } while (0)
}  // namespace

template <int NR, int ABL, bool DBG, bool SH = false>
__global__ __launch_bounds__(512) void conv3d_wino_pw_kernel(const WinoArgs a, int tiles_sp,
                                                             int total_tiles) {
  constexpr int kPNP = pw_pnp(SH), kPRS = pw_prs(SH);
  extern __shared__ __attribute__((aligned(16))) float lds_all[];
  float* R = lds_all;                                       // [2][600][8]
  float* V = R + 2 * kPRS;                                  // [2][6][16][16][8]
  float* S = V + 2 * kPVSZ;                                 // [4][NR*16][2] statistics scratch
  float* NT = S + 4 * NR * 16 * 2;                          // [2][cin_p] mean, rstd of the input image
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mrow = lane & 15, kq = lane >> 4;
  const int nk8 = a.cin_p >> 3, nb = a.cout_p16 >> 4;
  const int P = nk8;                                        // channel passes per tile (>= 3)

  // ---- this workgroup's tiles: XCD x = blockIdx % 8 owns one contiguous range of the linear
  // tile order (neighbouring tiles share halo pixels; the per-XCD L2s are not coherent with
  // each other), its workgroups take the tiles of that range round-robin
  const int xcd = blockIdx.x & 7, wi = blockIdx.x >> 3, per = gridDim.x >> 3;
  const int tq = total_tiles >> 3, tr = total_tiles & 7;
  const int t_start = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
  const int t_count = tq + (xcd < tr ? 1 : 0);
  const int n_my = t_count > wi ? (t_count - wi + per - 1) / per : 0;
  if (n_my == 0) return;
  const int bx_n = (a.W + kWTX - 1) / kWTX, by_n = (a.H + kWTY - 1) / kWTY;
  const WinoTiling tg = a.tiling;                             // (read from the kernel arguments once)
  auto decode = [&](int k) __attribute__((always_inline)) -> Tile {
    int L = t_start + wi + k * per;
    Tile t;
    if constexpr (SH) {
      // (shape-major inside a column-block group: a workgroup's stride walk changes shape at most twice per group)
      // (arithmetic, not a table: a scalar load here costs its latency AND an lgkmcnt(0) wait -- which also waits for
      //  the LDS operations in flight -- at every tile boundary of every wave: measured 0.65 against 0.57 ms per launch)
      const int S = a.N * tiles_sp;
      int g = 0;
      while (L >= S) { L -= S; ++g; }                       // (column-block groups: one to three; no division)
      const WinoTile w = wino_decode(tg, a.N, a.H, a.W, kPTZ, L);
      t.n = w.n; t.nb0 = g * NR; t.z0 = w.z0; t.y0 = w.y0; t.x0 = w.x0; t.lc = w.lc; t.hlim = w.hlim; t.wlim = w.wlim;
    } else {
      const int sp = L % tiles_sp; L /= tiles_sp;
      t.n = L % a.N;
      t.nb0 = (L / a.N) * NR;
      t.x0 = (sp % bx_n) * kWTX;
      t.y0 = ((sp / bx_n) % by_n) * kWTY;
      t.z0 = (sp / (bx_n * by_n)) * kPTZ;
      t.lc = 2; t.hlim = a.H; t.wlim = a.W;
    }
    return t;
  };

  if (wave >= 4) {
    // ===================================================================== staging waves
    const int ht = tid - 256, sw = wave - 4;
    if (JH_PW_PRIO) __builtin_amdgcn_s_setprio(JH_PW_PRIO);
    const bool dbg_on = DBG && a.dbg != nullptr && blockIdx.x == 0 && (wave == 4 || wave == 7);
    long long dbg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dbg_t = dbg_on ? __builtin_readcyclecounter() : 0;
    // Division of labour inside the staging role (every instruction of these waves competes
    // with the matrix wave of its SIMD for issue slots, so the work is spread by wave):
    //   wave 7 ("loader")     : global loads of the raw patch two to three passes ahead,
    //                           InstanceNorm(+act) on load, commit to the double-buffered R
    //   waves 4-6 (192 lanes) : the input transform of the 192 (z-slice, tile, channel quad)
    //                           items, R -> V, one pass ahead
    //   all four              : the epilogue of the previous tile, one z-slice each
    const bool loader = sw == 3;
    constexpr int ITER = (kPNP * 2 + 63) / 64;              // 1200 float4 items over the 64 loader lanes
    const int q = lane & 1;                                 // a lane's channel quad (all its items)
    float4 pf[ITER];                                        // patch of the pass in flight
    // ---- loader: patch addressing with ZERO per-load address arithmetic.  Item `it` of a lane is
    // patch pixel (lane + 64 it) / 2, channel quad q.  Its byte offset relative to the patch origin
    // (z0-1, y0-1, x0-1) does not depend on the tile: prel[it], computed once.  The tile's origin
    // goes into the base address of a buffer descriptor (scalar work), the channel pass into the
    // scalar offset of the load.  Pixels outside the volume (the convolution's zero padding)
    // get bit 31 set in their offset: beyond num_records, the buffer load returns 0 without
    // touching memory.  A pixel is outside exactly when it lies in the first patch plane of a
    // first tile, or past the volume's last plane in a last tile -- patch plane index > r, with r the
    // extent of the volume inside its last tile (the tile size when the extent is a multiple of it:
    // then only the last patch plane; round 4: any remainder, e.g. the 36^3 / 18^3 volumes of the
    // reference's shipped 72^3 grid): six constant lane masks, six scalar flags per tile.
    int prel[ITER];
    unsigned mz0 = 0, mz1 = 0, my0 = 0, my1 = 0, mx0 = 0, mx1 = 0, mtail = 0;
    const int rz = a.D % kPTZ ? a.D % kPTZ : kPTZ;
    // SH: the tables above are per block SHAPE (patch 10 x 10, 6 x 18 or 18 x 6 pixels per z-slice; the remainder of
    // the volume inside the last block of that shape's grid); rebuilt when a tile of another shape comes up -- at most
    // twice per column-block group, the tile list is shape-major.  TH / TW: voxels of the current shape's block.
    int shape_lc = -1, TH = kWTY, TW = kWTX;
    auto shape_tables = [&](auto pw_c, auto ph_c, int ry, int rx) __attribute__((always_inline)) {
      constexpr int PW = decltype(pw_c)::value, PH = decltype(ph_c)::value;
      mz0 = mz1 = my0 = my1 = mx0 = mx1 = mtail = 0;
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        const int idx = lane + it * 64;
        const int pix = idx >> 1;
        const int px = pix % PW, py = (pix / PW) % PH, pz = pix / (PW * PH);
        prel[it] = (((pz * a.H + py) * a.W + px) * a.cin_p + q * 4) * 4;
        mz0 |= (pz == 0) << it; mz1 |= (pz > rz) << it;
        my0 |= (py == 0) << it; my1 |= (py > ry) << it;
        mx0 |= (px == 0) << it; mx1 |= (px > rx) << it;
        mtail |= (idx >= kPPZ * PW * PH * 2) << it;         // (beyond the patch: never valid)
      }
    };
    auto set_shape = [&](int lc) __attribute__((always_inline)) {
      static_assert(!SH || ITER == kWinoShapeIter, "the host's shape tables cover 21 rounds");
      shape_lc = lc;
      TH = 32 >> lc; TW = 2 << lc;
      const int* tab = a.shape_tab + (lc - 1) * kWinoShapeWords + lane;
#pragma unroll
      for (int it = 0; it < ITER; ++it) prel[it] = tab[it * 64];
      mz0 = tab[ITER * 64]; mz1 = tab[(ITER + 1) * 64]; my0 = tab[(ITER + 2) * 64]; my1 = tab[(ITER + 3) * 64];
      mx0 = tab[(ITER + 4) * 64]; mx1 = tab[(ITER + 5) * 64]; mtail = tab[(ITER + 6) * 64];
    };
    if constexpr (!SH) {
      shape_tables(std::integral_constant<int, kWPX>{}, std::integral_constant<int, kWPY>{},
                   a.H % kWTY ? a.H % kWTY : kWTY, a.W % kWTX ? a.W % kWTX : kWTX);
    }
    int pvo[ITER];                                          // byte offsets of the patch to request
    unsigned pinv = 0;                                      // its invalid items
    __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, 0, 0x00020000);
    auto set_patch = [&](const Tile& t) __attribute__((always_inline)) {
      if constexpr (SH) {
        if (t.lc != shape_lc) set_shape(t.lc);
      }
      const long img_bytes = (long)a.D * a.H * a.W * a.cin_p * 4;
      const long org = ((long)((t.z0 - 1) * a.H + (t.y0 - 1)) * a.W + (t.x0 - 1)) * a.cin_p * 4;
      const char* base = reinterpret_cast<const char*>(a.x) + (long)t.n * img_bytes + org;
      prs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (int)(img_bytes - org), 0x00020000);
      pinv = mtail | (t.z0 == 0 ? mz0 : 0u) | (t.z0 + kPTZ >= a.D ? mz1 : 0u) | (t.y0 == 0 ? my0 : 0u) |
             (t.y0 + TH >= a.H ? my1 : 0u) | (t.x0 == 0 ? mx0 : 0u) | (t.x0 + TW >= a.W ? mx1 : 0u);
#pragma unroll
      for (int it = 0; it < ITER; ++it)
        pvo[it] = (int)(((pinv << (31 - it)) & 0x80000000u) | (unsigned)prel[it]);
    };
    typedef float bf32x4 __attribute__((ext_vector_type(4)));
    // (SH: a 4 x 4-tile block's patch needs 19 of the 21 rounds; running 19 for those blocks under a uniform branch
    //  measured the same -- 0.652 against 0.645 ms -- so every block runs the 21 rounds of the largest patch)
    auto issue = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        const bf32x4 v = __builtin_bit_cast(bf32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, pvo[it], c0 * 4, 0));
        pf[it] = make_float4(v[0], v[1], v[2], v[3]);
      }
    };
    // mean / rstd of image n (all channels) -> NT.  Written and read by the loader wave only
    // (program order inside one wave), so no barrier is involved.
    auto norm_table = [&](int n) __attribute__((always_inline)) {
      if (a.in_stats) {
        for (int c = lane; c < a.cin_p; c += 64) {
          const double* st = a.in_stats + ((size_t)n * a.cin_p + c) * kStatW;
          const double m = exact_read(st) * (double)a.in_inv;
          double var = exact_read(st + kLimbs) * (double)a.in_inv - m * m;
          if (var < 0.0) var = 0.0;
          NT[c] = (float)m;
          NT[a.cin_p + c] = (float)(1.0 / sqrt(var + 1e-5));
        }
      }
    };
    // Two fp32 lanes per VALU instruction (v_pk_add_f32 / v_pk_mul_f32): while the matrix waves
    // stream MFMAs this wave gets roughly one issue slot per MFMA, so its instruction COUNT is
    // what matters.
    // commit: InstanceNorm as one packed FMA per channel pair, x * rstd + (-mean * rstd); ReLU and
    // the zero padding of the NORMALISED tensor in one v_med3_f32 per value: med3(x, 0, M) with
    // M = +inf for pixels inside the volume, 0 outside (their loads returned 0, the FMA made that
    // -mean * rstd).  `inv` = invalid-item mask of the patch.
    // MODE: 0 = input already normalised (plain copy), 1 = InstanceNorm, 2 = + ReLU, 3 = + SiLU.
    // Dispatched ONCE per commit: no per-item control flow.
    auto commit_mode = [&](auto mode_c, unsigned inv, int c0, float* Rd) __attribute__((always_inline)) {
      constexpr int MODE = decltype(mode_c)::value;
      f32x2 nm[2], rs[2];
      if (MODE != 0) {
        const float4 m4 = *reinterpret_cast<const float4*>(NT + c0 + q * 4);
        const float4 r4 = *reinterpret_cast<const float4*>(NT + a.cin_p + c0 + q * 4);
        rs[0] = (f32x2){r4.x, r4.y}; rs[1] = (f32x2){r4.z, r4.w};
        nm[0] = (f32x2){-m4.x, -m4.y} * rs[0]; nm[1] = (f32x2){-m4.z, -m4.w} * rs[1];
      }
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        const float4 raw = pf[it];
        f32x2 lo = (f32x2){raw.x, raw.y}, hi = (f32x2){raw.z, raw.w};
        if (MODE != 0) {
          lo = __builtin_elementwise_fma(lo, rs[0], nm[0]);
          hi = __builtin_elementwise_fma(hi, rs[1], nm[1]);
          if (MODE == 2) {
            const float M = (inv >> it & 1) ? 0.f : __builtin_inff();
            lo = (f32x2){__builtin_amdgcn_fmed3f(lo.x, 0.f, M), __builtin_amdgcn_fmed3f(lo.y, 0.f, M)};
            hi = (f32x2){__builtin_amdgcn_fmed3f(hi.x, 0.f, M), __builtin_amdgcn_fmed3f(hi.y, 0.f, M)};
          } else {
            if (MODE == 3) {
              lo = (f32x2){silu_fast(lo.x), silu_fast(lo.y)};
              hi = (f32x2){silu_fast(hi.x), silu_fast(hi.y)};
            }
            const float m = (inv >> it & 1) ? 0.f : 1.f;
            lo *= (f32x2){m, m};
            hi *= (f32x2){m, m};
          }
        }
        // (R has room for the 16 items beyond the patch that the last iteration writes)
        *reinterpret_cast<float4*>(Rd + (lane + it * 64) * 4) = make_float4(lo.x, lo.y, hi.x, hi.y);   // [pix][q]
      }
    };
    const int cmode = !a.in_stats ? 0 : (a.in_act == ACT_RELU ? 2 : (a.in_act == ACT_SILU ? 3 : 1));
    auto commit = [&](unsigned inv, int c0, float* Rd) __attribute__((always_inline)) {
      if (cmode == 0) commit_mode(std::integral_constant<int, 0>{}, inv, c0, Rd);
      else if (cmode == 2) commit_mode(std::integral_constant<int, 2>{}, inv, c0, Rd);
      else if (cmode == 1) commit_mode(std::integral_constant<int, 1>{}, inv, c0, Rd);
      else commit_mode(std::integral_constant<int, 3>{}, inv, c0, Rd);
    };
    // input transform B^T d B of every (z-slice, tile, channel quad): 6 x 16 x 2 = 192 items
    auto transform = [&](const float* Rs, float* Vd, int lc) __attribute__((always_inline)) {
      {
        const int tile = (ht >> 1) & 15, pz = ht >> 5;
        // (tile -> row / column of the block; the patch of the block's shape)
        constexpr bool SHT = SH;
        const int ty = SHT ? tile >> lc : tile >> 2, tx = SHT ? tile & ((1 << lc) - 1) : tile & 3;
        const int PW = SHT ? (2 << lc) + 2 : kWPX, PH = SHT ? (32 >> lc) + 2 : kWPY;
        const float* rb = Rs + ((pz * PH + 2 * ty) * PW + 2 * tx) * 8 + q * 4;
        f32x2 tr[4][4][2];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          f32x2 d[4][2];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float4 v = *reinterpret_cast<const float4*>(rb + (r * PW + c) * 8);
            d[c][0] = (f32x2){v.x, v.y}; d[c][1] = (f32x2){v.z, v.w};
          }
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            tr[r][0][h] = d[0][h] - d[2][h];
            tr[r][1][h] = d[1][h] + d[2][h];
            tr[r][2][h] = d[2][h] - d[1][h];
            tr[r][3][h] = d[1][h] - d[3][h];
          }
        }
        // swizzle: channel quad q of tile rows 8..15 goes to slot q ^ 1
        float* vb = Vd + ((pz * 16) * 16 + tile) * 8 + ((q ^ (tile >> 3)) & 1) * 4;
#pragma unroll
        for (int fx = 0; fx < 4; ++fx) {
          f32x2 o[4][2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            o[0][h] = tr[0][fx][h] - tr[2][fx][h];
            o[1][h] = tr[1][fx][h] + tr[2][fx][h];
            o[2][h] = tr[2][fx][h] - tr[1][fx][h];
            o[3][h] = tr[1][fx][h] - tr[3][fx][h];
          }
#pragma unroll
          for (int fy2 = 0; fy2 < 4; ++fy2)
            *reinterpret_cast<float4*>(vb + (fy2 * 4 + fx) * 16 * 8) =
                make_float4(o[fy2][0].x, o[fy2][0].y, o[fy2][1].x, o[fy2][1].y);
        }
      }
    };
    // ---- epilogue of a finished tile from the two dump halves held in registers.
    // e[ox][w * NR + nr] = x-folded partial sums of matrix wave w (fy = w) for z-slice `sw`.
    // Along y across the matrix waves: out[0] = P0 + P1 + P2, out[1] = P1 - P2 - P3; then bias,
    // statistics, 4x4 quad transpose (a lane holds tiles (ty = kq, tx = 0..3) of channel mrow;
    // afterwards tile tx = lane & 3 of channels (mrow & ~3) .. + 3), 16-byte stores.
    // The x phase ox = 0 is finished right at the tile boundary (the matrix waves wait for it at
    // Bd3: a few hundred cycles), ox = 1 during the next tile's first pass, so only ONE dump half
    // (12 float4 for NR = 3) is ever held in registers.
    float4 e[4 * NR];
    float s1[NR], s2[NR];                                   // statistics of the tile being finished
    const int jq = lane & 3;
    auto finish_half = [&](const Tile& t, int ox) __attribute__((always_inline)) {
      constexpr bool SHF = SH;
      // a lane holds tiles 4 kq + r (r = 0..3) of the block: tile -> (row tile >> lc, column tile & cmask); outputs at
      // y >= hlim or x >= wlim belong to another block (the strips) or lie outside the volume.  (Kept as arithmetic on
      // the spot: the ten lane constants per shape that would replace the shifts cost registers the kernel does not
      // have -- 0.68 against 0.60 ms per launch.)
      const int lc = SHF ? t.lc : 2, cmask = (1 << lc) - 1;
      const int hlim = SHF ? t.hlim : a.H, wlim = SHF ? t.wlim : a.W;
      float* yb = a.y + (size_t)t.n * a.D * a.H * a.W * a.cout_p;
      const int oz = t.z0 + sw;
#pragma unroll
      for (int nr = 0; nr < NR; ++nr) {
        const int ch = (t.nb0 + nr) * 16 + mrow;
        const bool ch_ok = ch < a.cout_p;
        const float bvl = (a.bias && ch < a.cout_p16) ? a.bias[ch] : 0.f;
        if (ox == 0) { s1[nr] = 0.f; s2[nr] = 0.f; }
        const float4 p0 = e[0 * NR + nr], p1 = e[1 * NR + nr], p2 = e[2 * NR + nr], p3 = e[3 * NR + nr];
#pragma unroll
        for (int oy = 0; oy < 2; ++oy) {
          float v[4];
          if (oy == 0) {
            v[0] = p0.x + p1.x + p2.x; v[1] = p0.y + p1.y + p2.y;
            v[2] = p0.z + p1.z + p2.z; v[3] = p0.w + p1.w + p2.w;
          } else {
            v[0] = p1.x - p2.x - p3.x; v[1] = p1.y - p2.y - p3.y;
            v[2] = p1.z - p2.z - p3.z; v[3] = p1.w - p2.w - p3.w;
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] += bvl;
            const int tl = 4 * kq + r;
            if (ch_ok && oz < a.D && t.y0 + 2 * (tl >> lc) + oy < hlim && t.x0 + 2 * (tl & cmask) + ox < wlim) {
              s1[nr] += v[r];
              s2[nr] += v[r] * v[r];
            }
          }
          {
            float x, y;
            x = (jq & 1) ? v[0] : v[1]; y = quad_xor1(x); if (jq & 1) v[0] = y; else v[1] = y;
            x = (jq & 1) ? v[2] : v[3]; y = quad_xor1(x); if (jq & 1) v[2] = y; else v[3] = y;
            x = (jq & 2) ? v[0] : v[2]; y = quad_xor2(x); if (jq & 2) v[0] = y; else v[2] = y;
            x = (jq & 2) ? v[1] : v[3]; y = quad_xor2(x); if (jq & 2) v[1] = y; else v[3] = y;
          }
          const int tj = 4 * kq + jq;                       // (after the transpose: tile 4 kq + jq)
          const int yy = t.y0 + 2 * (tj >> lc) + oy, xx = t.x0 + 2 * (tj & cmask) + ox;
          const int c0 = (t.nb0 + nr) * 16 + (mrow & ~3);
          if (c0 < a.cout_p && oz < a.D && yy < hlim && xx < wlim)
            *reinterpret_cast<float4*>(yb + ((size_t)(oz * a.H + yy) * a.W + xx) * a.cout_p + c0) =
                make_float4(v[0], v[1], v[2], v[3]);
        }
        if (ox == 1 && a.stats) {
          float t1 = s1[nr], t2 = s2[nr];
          t1 = sum_xor16(t1); t2 = sum_xor16(t2);
          t1 = sum_xor32(t1); t2 = sum_xor32(t2);
          if (kq == 0) {
            S[(sw * NR * 16 + nr * 16 + mrow) * 2 + 0] = t1;
            S[(sw * NR * 16 + nr * 16 + mrow) * 2 + 1] = t2;
          }
        }
      }
    };
    auto atomics = [&](const Tile& t) __attribute__((always_inline)) {                     // (one barrier after finish())
      if (a.stats && ht < NR * 16) {
        const int ch = t.nb0 * 16 + ht;
        if (ch < a.cout_p) {
          float t1 = 0.f, t2 = 0.f;
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            t1 += S[(w * NR * 16 + ht) * 2 + 0];
            t2 += S[(w * NR * 16 + ht) * 2 + 1];
          }
          stat_add(a.stats + ((size_t)t.n * a.cout_p + ch) * kStatW, t1, t2);
        }
      }
    };
    auto read_dump = [&](const float* D) __attribute__((always_inline)) {
      const float4* D4 = reinterpret_cast<const float4*>(D);
#pragma unroll
      for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) e[w * NR + nr] = D4[((w * kPTZ + sw) * NR + nr) * 64 + lane];
    };

    // Schedule, with g = global pass index (tile k, pass p): during matrix pass g the loader
    // commits pass g+2 into R[g & 1] and then requests pass g+3, the transform waves turn
    // R[(g+1) & 1] into V[(g+1) & 1].  R[g & 1] was last read by transform(g) during pass g-1,
    // V[(g+1) & 1] last by the matrix waves during pass g-1; the end-of-pass barrier separates
    // all of it.  pi / pm: the patches the next request / commit belong to (P >= 3: never more
    // than one tile ahead); pi moves to the next tile at p = P-3, pm at p = P-2.
    float* R0 = R;
    float* R1 = R + kPRS;
    // The two sub-roles run the same barrier sequence from two separate instantiations of this
    // body, so that the loader's patch registers and the transform's temporaries never share a
    // live range.
    auto run = [&](auto is_loader) __attribute__((always_inline)) {
      constexpr bool LOADER = decltype(is_loader)::value;
      Tile cur = decode(0), prev = cur, nxt = cur;
      unsigned pm = 0;                                      // validity mask of the patch to commit
      if constexpr (LOADER) {
        set_patch(cur);
        pm = pinv;
        issue(0);
        norm_table(cur.n);
        commit(pm, 0, R0);
        issue(8);
      }
      wg_barrier();                                         // P0: R[0] = pass 0
      if constexpr (LOADER) {
        commit(pm, 8, R1);
        issue(16);                                          // P >= 3
      } else {
        transform(R0, V, cur.lc);
      }
      wg_barrier();                                         // P1: V[0], R[1] complete
      wg_barrier();                                         // P2 (kept: the matrix waves count three)
      int g = 0;                                            // global pass index of (k, p = 0)
      for (int k = 0; k < n_my; ++k) {
        const bool has_next = k + 1 < n_my;
        if (has_next) nxt = decode(k + 1);
        for (int p = 0; p < P; ++p) {
          JH_T(7);
          if (p == 0 && k > 0 && ABL != 1) finish_half(prev, 1);
          JH_T(0);
          if constexpr (LOADER) {
            // pf holds pass g+2 (requested one pass ago)
            if (p == P - 2 && has_next) {
              pm = pinv;
              if (nxt.n != cur.n) norm_table(nxt.n);        // all commits of this tile are done
            }
            if ((p + 2 < P || has_next) && ABL != 2)
              commit(pm, (p + 2 < P ? p + 2 : p + 2 - P) * 8, (g + p) & 1 ? R1 : R0);
            JH_T(1);
            if (p == P - 3 && has_next) set_patch(nxt);
            if ((p + 3 < P || has_next) && ABL != 2) issue((p + 3 < P ? p + 3 : p + 3 - P) * 8);
            JH_T(2);
          } else {
            if ((p + 1 < P || has_next) && ABL != 2)
              transform((g + p + 1) & 1 ? R1 : R0, V + ((g + p + 1) & 1) * kPVSZ, p + 1 < P ? cur.lc : nxt.lc);
            JH_T(4);
          }
          JH_T(3);
          wg_barrier();                                     // Be (also: S of finish_half() complete)
          JH_T(5);
          if (p == 0 && k > 0) atomics(prev);
        }
        g += P;
        const float* D = V + ((g + 1) & 1) * kPVSZ;         // buffer of the pass just consumed
        wg_barrier();                                       // Bd1: dump(ox 0) complete
        read_dump(D);
        wg_barrier();                                       // Bd2: ... and read
        if (ABL != 1) finish_half(cur, 0);
        wg_barrier();                                       // Bd3: dump(ox 1) complete
        read_dump(D);
        wg_barrier();                                       // Bd4: every staging wave has read it (the
                                                            // transform waves overwrite D in the next pass)
        prev = cur; cur = nxt;
        JH_T(6);
      }
      if (dbg_on && lane == 0) for (int i = 0; i < 8; ++i) if (dbg_acc[i]) a.dbg[(LOADER ? 16 : 0) + i] = dbg_acc[i];
      finish_half(prev, 1);
      __syncthreads();                                      // Bf
      atomics(prev);
    };
    if (loader) run(std::true_type{}); else run(std::false_type{});
    return;
  }

  // ======================================================================= matrix waves
  const bool dbg_on = DBG && a.dbg != nullptr && blockIdx.x == 0 && wave == 0;
  long long dbg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dbg_t = dbg_on ? __builtin_readcyclecounter() : 0;
  f32x4 acc[4][kPTZ][NR];
  const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.u), 0, 16 * 3 * nk8 * nb * 512, 0x00020000);
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  int bvo[NR];
  auto set_group = [&](int nb0) __attribute__((always_inline)) {
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) bvo[nr] = (min(nb0 + nr, nb - 1) * 64 + lane) * 8;
  };
  auto load_b = [&](float2 (&bd)[NR], int f, int dz, int kk) {
    const int so = (((f * 3 + dz) * nk8 + kk) * nb) * 512;
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
      const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(urs, bvo[nr], so, 0);
      bd[nr] = make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
    }
  };
  // A operand: row = tile (mrow), k-quad kq; swizzled like the transform writes it
  const int aoff = mrow * 4 + (kq ^ ((mrow >> 3) << 1));
  float2 b[3][NR], av[kPPZ], avn[kPPZ];
  Tile cur = decode(0);
  set_group(cur.nb0);
#pragma unroll
  for (int dz = 0; dz < 3; ++dz) load_b(b[dz], wave * 4, dz, 0);
  wg_barrier();                                             // P0
  wg_barrier();                                             // P1
  wg_barrier();                                             // P2: V[0] complete
  {
    const float2* V2 = reinterpret_cast<const float2*>(V);
#pragma unroll
    for (int pz = 0; pz < kPPZ; ++pz) av[pz] = V2[((pz * 16 + wave * 4) * 16) * 4 + aoff];
  }
#pragma unroll
  for (int fi = 0; fi < 4; ++fi)
#pragma unroll
    for (int mr = 0; mr < kPTZ; ++mr)
#pragma unroll
      for (int nr = 0; nr < NR; ++nr) acc[fi][mr][nr] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int g = 0;
  for (int k = 0; k < n_my; ++k) {
    const bool has_next = k + 1 < n_my;
    const Tile nxt = has_next ? decode(k + 1) : cur;
    for (int p = 0; p < P; ++p) {
      const float2* V2 = reinterpret_cast<const float2*>(V + ((g + p) & 1) * kPVSZ);
      const int pnx = p + 1 < P ? p + 1 : 0;                // (wraps into the next tile)
#pragma unroll
      for (int fi = 0; fi < 4; ++fi) {
        const int f = wave * 4 + fi;
        const int nf = fi < 3 ? f + 1 : wave * 4, nk = fi < 3 ? p : pnx;
#pragma unroll
        for (int dz = 0; dz < 3; ++dz) {
          if (ABL != 3) {
#pragma unroll
          for (int mr = 0; mr < kPTZ; ++mr)
#pragma unroll
            for (int nr = 0; nr < NR; ++nr)
              acc[fi][mr][nr] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mr + dz].x, b[dz][nr].x, acc[fi][mr][nr], 0, 0, 0);
#pragma unroll
          for (int mr = 0; mr < kPTZ; ++mr)
#pragma unroll
            for (int nr = 0; nr < NR; ++nr)
              acc[fi][mr][nr] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mr + dz].y, b[dz][nr].y, acc[fi][mr][nr], 0, 0, 0);
          } else {
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) acc[fi][dz][nr][0] += av[dz].x * b[dz][nr].x + av[dz + 3].y * b[dz][nr].y;
          }
          __builtin_amdgcn_sched_barrier(0);
          load_b(b[dz], nf, dz, nk);
          if (fi < 3) {
            avn[2 * dz] = V2[(((2 * dz) * 16 + nf) * 16) * 4 + aoff];
            avn[2 * dz + 1] = V2[(((2 * dz + 1) * 16 + nf) * 16) * 4 + aoff];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (fi < 3) {
#pragma unroll
          for (int pz = 0; pz < kPPZ; ++pz) av[pz] = avn[pz];
        }
        if (fi == 3) { JH_T(0); wg_barrier(); JH_T(2); }    // Be: one barrier per channel pass
        if (fi == 3 && p + 1 < P) {                         // V[(g+p+1) & 1] is complete now
          const float2* Vn = reinterpret_cast<const float2*>(V + ((g + p + 1) & 1) * kPVSZ);
#pragma unroll
          for (int pz = 0; pz < kPPZ; ++pz) av[pz] = Vn[((pz * 16 + wave * 4) * 16) * 4 + aoff];
        }
      }
    }
    g += P;
    // ---- fold fx (A^T along x: r0 = M0 + M1 + M2, r1 = M1 - M2 - M3) and hand the partial
    // sums to the staging waves through the V buffer of the pass just consumed
    float4* D4 = reinterpret_cast<float4*>(V + ((g + 1) & 1) * kPVSZ);
#pragma unroll
    for (int mr = 0; mr < kPTZ; ++mr)
#pragma unroll
      for (int nr = 0; nr < NR; ++nr) {
        const f32x4 m0 = acc[0][mr][nr], m1 = acc[1][mr][nr], m2 = acc[2][mr][nr];
        D4[((wave * kPTZ + mr) * NR + nr) * 64 + lane] =
            make_float4(m0[0] + m1[0] + m2[0], m0[1] + m1[1] + m2[1], m0[2] + m1[2] + m2[2], m0[3] + m1[3] + m2[3]);
      }
    if (has_next) {                                         // V[g & 1] (next tile, pass 0) has been complete
      const float2* Vn = reinterpret_cast<const float2*>(V + (g & 1) * kPVSZ);   // since the last barrier
#pragma unroll
      for (int pz = 0; pz < kPPZ; ++pz) av[pz] = Vn[((pz * 16 + wave * 4) * 16) * 4 + aoff];
    }
    wg_barrier();                                           // Bd1
    wg_barrier();                                           // Bd2: the staging waves hold ox 0
#pragma unroll
    for (int mr = 0; mr < kPTZ; ++mr)
#pragma unroll
      for (int nr = 0; nr < NR; ++nr) {
        const f32x4 m1 = acc[1][mr][nr], m2 = acc[2][mr][nr], m3 = acc[3][mr][nr];
        D4[((wave * kPTZ + mr) * NR + nr) * 64 + lane] =
            make_float4(m1[0] - m2[0] - m3[0], m1[1] - m2[1] - m3[1], m1[2] - m2[2] - m3[2], m1[3] - m2[3] - m3[3]);
      }
#pragma unroll
    for (int fi = 0; fi < 4; ++fi)
#pragma unroll
      for (int mr = 0; mr < kPTZ; ++mr)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) acc[fi][mr][nr] = (f32x4){0.f, 0.f, 0.f, 0.f};
    wg_barrier();                                           // Bd3
    wg_barrier();                                           // Bd4
    JH_T(3);
    if (has_next && nxt.nb0 != cur.nb0) {                   // other column-block group: the weights
      set_group(nxt.nb0);                                   // requested ahead were the wrong ones
#pragma unroll
      for (int dz = 0; dz < 3; ++dz) load_b(b[dz], wave * 4, dz, 0);
    }
    cur = nxt;
  }
  if (dbg_on && lane == 0) for (int i = 0; i < 8; ++i) a.dbg[8 + i] = dbg_acc[i];
  __syncthreads();                                          // Bf
}

template <int NR, int ABL = 0, bool DBG = false, bool SH = false>
static int launch_pw_nr(const WinoArgs& a, int grid, size_t lds, int tiles_sp, int total, hipStream_t s) {
  auto kern = conv3d_wino_pw_kernel<NR, ABL, DBG, SH>;
  static bool big = false;
  if (!big) {
    JH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    big = true;
  }
  static long long* dbg = nullptr;
  WinoArgs b = a;
  if (DBG) {
    if (!dbg) JH_CHECK_HIP(hipMalloc(&dbg, 24 * sizeof(long long)));
    JH_CHECK_HIP(hipMemsetAsync(dbg, 0, 24 * sizeof(long long), s));
    b.dbg = dbg;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, b, tiles_sp, total);
  JH_CHECK_HIP(hipGetLastError());
  if (b.dbg) {
    long long h[24];
    JH_CHECK_HIP(hipStreamSynchronize(s));
    JH_CHECK_HIP(hipMemcpy(h, dbg, sizeof h, hipMemcpyDeviceToHost));
    fprintf(stderr, "[pw dbg cin_p=%d tiles=%d] transform wave: finish %lld transform %lld waitBm %lld waitBe %lld dump %lld "
            "other %lld | loader: finish %lld commit %lld issue %lld waitBm %lld waitBe %lld dump %lld other %lld | matrix: mfma %lld "
            "waitBm %lld waitBe %lld dump %lld\n", a.cin_p, total, h[0], h[4], h[3], h[5], h[6], h[7],
            h[16], h[17], h[18], h[19], h[21], h[22], h[23], h[8], h[9], h[10], h[11]);
  }
  return 0;
}

// Returns -1 when this launch is not worth (or not able) to run persistently: fewer than two
// tiles per CU, or a single channel pass -- the caller falls back to conv3d_wino_kernel.
int launch_conv3d_wino_pw(const WinoArgs& a, int nr, hipStream_t s) {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    cus = prop.multiProcessorCount / 8 * 8;
    if (cus < 8) return -1;
  }
  const int nb = a.cout_p16 / 16;
  const bool sh = a.tiling.shaped != 0;
  const int tiles_sp = sh ? a.tiling.slabs * wino_blocks_per_slab(a.tiling)
                          : ((a.D + kPTZ - 1) / kPTZ) * ((a.H + kWTY - 1) / kWTY) * ((a.W + kWTX - 1) / kWTX);
  const int groups = (nb + nr - 1) / nr;
  const long total = (long)tiles_sp * groups * a.N;
  if (a.cin_p < 24 || total < 2L * cus || total > (1L << 30)) return -1;       // P >= 3
  if ((long)a.D * a.H * a.W * a.cin_p * 4 + (long)(a.H + 1) * a.W * a.cin_p * 4 + 64 >= (1L << 31)) return -1;
  JH_REQUIRE((size_t)4 * kPTZ * nr * 64 * 4 <= (size_t)kPVSZ, "wino (persistent) dump buffer");
  const size_t lds = (size_t)(2 * pw_prs(sh) + 2 * kPVSZ + 4 * nr * 16 * 2 + 2 * a.cin_p) * sizeof(float);
  JH_REQUIRE(lds <= 160 * 1024, "wino (persistent) LDS");
  if (sh) {       // (the volumes with remainder strips: no experiment variants)
    JH_REQUIRE(a.shape_tab != nullptr, "wino (persistent) shape tables");
    if (nr == 3) return launch_pw_nr<3, 0, false, true>(a, cus, lds, tiles_sp, (int)total, s);
    if (nr == 2) return launch_pw_nr<2, 0, false, true>(a, cus, lds, tiles_sp, (int)total, s);
    return launch_pw_nr<1, 0, false, true>(a, cus, lds, tiles_sp, (int)total, s);
  }
  const int abl = JH_ENV_KNOB("JH_WS_ABL");
  if (nr == 3 && abl == 1) return launch_pw_nr<3, 1>(a, cus, lds, tiles_sp, (int)total, s);
  if (nr == 3 && abl == 2) return launch_pw_nr<3, 2>(a, cus, lds, tiles_sp, (int)total, s);
  if (nr == 3 && abl == 3) return launch_pw_nr<3, 3>(a, cus, lds, tiles_sp, (int)total, s);
  if (nr == 3 && JH_ENV_KNOB("JH_WINO_DBG") > 0) return launch_pw_nr<3, 0, true>(a, cus, lds, tiles_sp, (int)total, s);
  if (nr == 3) return launch_pw_nr<3>(a, cus, lds, tiles_sp, (int)total, s);
  if (nr == 2) return launch_pw_nr<2>(a, cus, lds, tiles_sp, (int)total, s);
  return launch_pw_nr<1>(a, cus, lds, tiles_sp, (int)total, s);
}

}  // namespace jh
