// ============================================================================
// orb_extractor.hip -- MI355X (gfx950) ORB extractor behind the C ABI of
// include/orbslam_hip.h.  Drop-in for ORB_SLAM2::ORBextractor
// (reference include/ORBextractor.h:45-111, src/ORBextractor.cc).
//
// Pipeline for a batch of B same-sized frames, everything resident in HBM:
//   k_resize        x (L-1)  level l from level l-1, fixed-point bilinear          (E2, :1107-1132)
//   k_fast_cells    x 1      one wave per ~30x30 cell: FAST-9 score map in LDS,
//                            3x3 NMS inside the cell, 20 -> 7 threshold fallback,
//                            ordered (row-major) compaction into the cell's slot   (E3, :789-829)
//   k_octree        x 1      one workgroup per (frame, level): DistributeOctTree in
//                            its level-synchronous array form (tests/octree_twin.py) (E4, :539-763)
//   k_blur7         x 1      separable 7x7 Gaussian, fixed-point taps, reflect-101  (E7, :1085-1086)
//   k_describe      x 1      one 32-lane half-wave per keypoint: intensity-centroid angle on the
//                            un-blurred level, rotated BRIEF-256 on the blurred one,
//                            cv::KeyPoint record                                     (E5,E6,E8,E9)
// All integer stages are bit-exact by construction; the two float stages
// (fastAtan2, rotated sampling coordinates) use explicit round-to-nearest ops
// with no FMA contraction so they equal the canonical CPU definition
// (SURVEY F11 / DESIGN.md).  No CPU fallback exists in this file.
// ============================================================================
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <atomic>
#include <vector>
#include <algorithm>
#include <new>
#include <type_traits>

#include "common.h"
#include "orb_pattern_data.h"

namespace orbhip {

static const int PATCH_SIZE = 31, HALF_PATCH = 15, EDGE_THRESHOLD = 19;
static const int MAX_LEVELS = 16;
static const int MAX_INI = 64;            // initial octree nodes per level (round(W/H))
static const int KEYCAP_MAX = 1 << 23;     // the dense candidate array of a (frame, level) is sized for its theoretical worst case (cells x
                                           // in-cell NMS density 1/4); this bound only keeps the 24-bit candidate index of the octree's
                                           // best-key word valid (a 4095 x 4095 level has at most 4.2 M).  ORBHIP_KEYCAP lowers it (test hook)

struct LevelDev {
  int w, h;
  int pitch;            // bytes per row of the un-blurred level (level 0: the caller's stride)
  int bpitch;           // bytes per row of the blurred level
  long long pyr_off;    // byte offset inside one frame's pyramid block (levels >= 1)
  long long blur_off;   // byte offset inside one frame's blurred block
  int minBX, minBY, winW, winH;   // detection window origin and size (maxBorder - minBorder)
  int cell_begin, ncells;
  int quota;
  int nIni; float hX;
  int ini_x[MAX_INI + 1];
  float scale; float patch;
  int kcap; int key_off;          // dense key capacity / offset (in keys) inside one frame's key block
  int dblk_begin, dblk_count;     // k_describe: first workgroup of this level / number of workgroups (level capacity / DESC_WPB)
};

struct GeomDev {
  int nlevels, ncells_total, cell_cap, sel_cap, keys_per_frame, desc_blocks;
  int tile_w, tile_h, tile_pitch;       // FAST LDS tile (max cell incl. apron)
  int node_cap, max_cells_level;
#ifdef ORBHIP_OCT_LEVEL_EXPERIMENT
  int oct_level_mask;
#endif
  long long pyr_frame_bytes, blur_frame_bytes;
  LevelDev lv[MAX_LEVELS];
};

struct alignas(16) CellDesc { short level, x0, y0, x1, y1, offx, offy, pad; };   // 16-byte aligned: read with one scalar load
struct alignas(8) BlurTile { short level, tx, ty, pad; };

// ---------------------------------------------------------------------------- device helpers
__device__ __forceinline__ const uint8_t* level_ptr(const GeomDev& G, int l, int f, const uint8_t* img0,
                                                    long long img_frame_bytes, const uint8_t* pyr) {
  return l == 0 ? img0 + (long long)f * img_frame_bytes
                : pyr + (long long)f * G.pyr_frame_bytes + G.lv[l].pyr_off;
}

// ---- wave-level integer sum / inclusive scan on DPP row operations (VALU only; __shfl goes through the LDS crossbar, ~100
// cycles of dependent latency per step) -------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xF, true); }
__device__ __forceinline__ int wave_sum_i32(int v) {              // total in every lane (readlane of lane 63)
  v += dpp_i32<0xB1, 0xF>(v);       // quad_perm [1,0,3,2]
  v += dpp_i32<0x4E, 0xF>(v);       // quad_perm [2,3,0,1]
  v += dpp_i32<0x141, 0xF>(v);      // row_half_mirror
  v += dpp_i32<0x140, 0xF>(v);      // row_mirror
  v += dpp_i32<0x142, 0xA>(v);      // row_bcast15 -> rows 1, 3
  v += dpp_i32<0x143, 0xC>(v);      // row_bcast31 -> rows 2, 3
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_incl_scan_i32(int v) {        // Hillis-Steele inside each row of 16, then the row totals
  v += dpp_i32<0x111, 0xF>(v);      // row_shr:1
  v += dpp_i32<0x112, 0xF>(v);      // row_shr:2
  v += dpp_i32<0x114, 0xF>(v);      // row_shr:4
  v += dpp_i32<0x118, 0xF>(v);      // row_shr:8
  v += dpp_i32<0x142, 0xA>(v);      // row_bcast15 -> rows 1, 3
  v += dpp_i32<0x143, 0xC>(v);      // row_bcast31 -> rows 2, 3
  return v;
}

// ---------------------------------------------------------------------------- k_resize
// xtab[dx] = {sx | a0 << 16, a0 | a1 << 16}: source column and the two fixed-point weights (0..2048) of output column dx.
// Each thread produces 4 output pixels; their <= 7 distinct source columns per row come from three
// aligned dwords when the source pitch allows (every level >= 1, and level 0 when stride % 4 == 0).
typedef unsigned short ushort2_rs __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int byte_of(uint32_t w0, uint32_t w1, uint32_t w2, int k) {   // byte k of the 12-byte window
  const uint32_t w = k < 4 ? w0 : (k < 8 ? w1 : w2);
  return (w >> (8 * (k & 3))) & 255;
}
#define RS_ROWS 2      // output rows per thread (measured: 1 row 0.375 ms, 2 rows 0.328, 3 rows 0.333, 4 rows 0.366, 8 rows 0.478): the x tables are read once per thread and the rows' source loads are in flight together
__global__ __launch_bounds__(256) void k_resize(const uint8_t* __restrict__ src, int spitch, long long sframe,
                                                int sw, int sh, uint8_t* __restrict__ dst, int dpitch,
                                                long long dframe, int dw, int dh, const uint2* __restrict__ xtab,
                                                const int* __restrict__ yofs, const short* __restrict__ ibeta) {
  const int x4 = (blockIdx.x * 64 + threadIdx.x) * 4;
  const int y0 = (blockIdx.y * 4 + threadIdx.y) * RS_ROWS;
  const int f = blockIdx.z;
  if (x4 >= dw || y0 >= dh) return;
  int sx[4], a0[4], a1[4];
  uint32_t wp[4];                                   // a0 | a1 << 16
  {
    // (the table is padded to a multiple of 4 columns with copies of the last one: two 16-byte requests instead of four clamped 8-byte ones)
    const uint4 t01 = ((const uint4*)xtab)[x4 >> 1], t23 = ((const uint4*)xtab)[(x4 >> 1) + 1];
    const uint32_t tx[4] = {t01.x, t01.z, t23.x, t23.z}, ty[4] = {t01.y, t01.w, t23.y, t23.w};
#pragma unroll
    for (int j = 0; j < 4; j++) { sx[j] = tx[j] & 0xFFFF; a0[j] = (int)(ty[j] & 0xFFFF); a1[j] = (int)(ty[j] >> 16); wp[j] = ty[j]; }
  }
  const int base = sx[0] & ~3;
  const bool window = (sx[3] + 1 - base < 12);
  // ---- phase 1: every row's source dwords are requested before any arithmetic (one thread used to do one row: two
  // dependent global round trips around ~100 VALU instructions; PMC: 49 % VALU-busy at 3.8 waves per SIMD) ----
  uint32_t P[RS_ROWS][3], Q[RS_ROWS][3];
  int bb0[RS_ROWS], bb1[RS_ROWS], mode[RS_ROWS];      // mode 0: row outside, 1: dword window in P / Q, 2: byte path
  const uint8_t* R0[RS_ROWS]; const uint8_t* R1[RS_ROWS];
#pragma unroll
  for (int r = 0; r < RS_ROWS; r++) {
    const int y = __builtin_amdgcn_readfirstlane(y0 + r);   // a wave is one row group (blockDim.x = 64): the row tables come by scalar loads
    mode[r] = 0;
    if (y < dh) {
      const int sy = yofs[y];
      const int sy0 = min(max(sy, 0), sh - 1), sy1 = min(max(sy + 1, 0), sh - 1);
      const uint8_t* S0 = src + (long long)f * sframe + (long long)sy0 * spitch;
      const uint8_t* S1 = src + (long long)f * sframe + (long long)sy1 * spitch;
      R0[r] = S0; R1[r] = S1;
      bb0[r] = ibeta[2 * y]; bb1[r] = ibeta[2 * y + 1];
      const bool fast = ((spitch & 3) == 0) && ((((size_t)S0) & 3) == 0) && (base + 12 <= spitch) && window;
      // rows of any alignment (level 1 reads the caller's frames, e.g. a 1241-byte stride): four ALIGNED dwords re-cut with
      // v_alignbyte give the same 12-byte window; base + 16 <= sw keeps every byte read inside the source row
      const bool fast_unaligned = !fast && window && (base + 16 <= sw) && (base >= 4 || sy0 > 0 || f > 0);   // (the <= 3 bytes read before S0 + base stay inside the buffer)
      if (fast) {
        P[r][0] = *(const uint32_t*)(S0 + base); P[r][1] = *(const uint32_t*)(S0 + base + 4); P[r][2] = *(const uint32_t*)(S0 + base + 8);
        Q[r][0] = *(const uint32_t*)(S1 + base); Q[r][1] = *(const uint32_t*)(S1 + base + 4); Q[r][2] = *(const uint32_t*)(S1 + base + 8);
        mode[r] = 1;
      } else if (fast_unaligned) {
        const uint8_t* a = S0 + base; const uint8_t* b = S1 + base;
        const uint32_t sa = (uint32_t)((size_t)a & 3), sb = (uint32_t)((size_t)b & 3);
        const uint32_t* pa = (const uint32_t*)(a - sa); const uint32_t* pb = (const uint32_t*)(b - sb);
        const uint32_t a0d = pa[0], a1d = pa[1], a2d = pa[2], a3d = pa[3], b0d = pb[0], b1d = pb[1], b2d = pb[2], b3d = pb[3];
        P[r][0] = __builtin_amdgcn_alignbyte(a1d, a0d, sa); P[r][1] = __builtin_amdgcn_alignbyte(a2d, a1d, sa); P[r][2] = __builtin_amdgcn_alignbyte(a3d, a2d, sa);
        Q[r][0] = __builtin_amdgcn_alignbyte(b1d, b0d, sb); Q[r][1] = __builtin_amdgcn_alignbyte(b2d, b1d, sb); Q[r][2] = __builtin_amdgcn_alignbyte(b3d, b2d, sb);
        mode[r] = 1;
      } else {
        mode[r] = 2;
      }
    }
  }
  // ---- phase 2: interpolate and store ----
#pragma unroll
  for (int r = 0; r < RS_ROWS; r++) {
    if (mode[r] == 0) continue;
    const int b0 = bb0[r], b1 = bb1[r];
    uint32_t out = 0;
    if (mode[r] == 1) {
      const uint32_t p0 = P[r][0], p1 = P[r][1], p2 = P[r][2], q0 = Q[r][0], q1 = Q[r][1], q2 = Q[r][2];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (x4 + j < dw) {
          // bytes k0, k0 + 1 of the 12-byte window as one u16 pair, times (a0, a1) with ONE v_dot2_u32_u16 per row.  At the right
          // border the reference clamps the second column and its weight a1 is 0, so the byte after the row never matters.
          const int k0 = sx[j] - base;
          const bool hi = k0 >= 4, hi2 = k0 >= 8;
          const uint32_t lo0 = hi2 ? p2 : (hi ? p1 : p0), up0 = hi2 ? 0u : (hi ? p2 : p1);
          const uint32_t lo1 = hi2 ? q2 : (hi ? q1 : q0), up1 = hi2 ? 0u : (hi ? q2 : q1);
          const uint32_t w0 = __builtin_amdgcn_alignbyte(up0, lo0, (uint32_t)(k0 & 3)), w1 = __builtin_amdgcn_alignbyte(up1, lo1, (uint32_t)(k0 & 3));
          const int H0 = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(ushort2_rs, __builtin_amdgcn_perm(0u, w0, 0x0c010c00u)), __builtin_bit_cast(ushort2_rs, wp[j]), 0u, false);
          const int H1 = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(ushort2_rs, __builtin_amdgcn_perm(0u, w1, 0x0c010c00u)), __builtin_bit_cast(ushort2_rs, wp[j]), 0u, false);
          const int v = (((b0 * (H0 >> 4)) >> 16) + ((b1 * (H1 >> 4)) >> 16) + 2) >> 2;
          out |= (uint32_t)(v & 255) << (8 * j);
        }
      }
    } else {
      const uint8_t* S0 = R0[r]; const uint8_t* S1 = R1[r];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (x4 + j < dw) {
          const int s0 = sx[j], s1 = min(sx[j] + 1, sw - 1);
          const int H0 = S0[s0] * a0[j] + S0[s1] * a1[j];
          const int H1 = S1[s0] * a0[j] + S1[s1] * a1[j];
          const int v = (((b0 * (H0 >> 4)) >> 16) + ((b1 * (H1 >> 4)) >> 16) + 2) >> 2;
          out |= (uint32_t)(v & 255) << (8 * j);
        }
      }
    }
    *(uint32_t*)(dst + (long long)f * dframe + (long long)(y0 + r) * dpitch + x4) = out;
  }
}

// ---------------------------------------------------------------------------- k_pyr_cone (single-frame latency path)
// The pyramid is a CHAIN: level l is resized from level l-1 (src/ORBextractor.cc:1107-1132), and one launch per level costs a
// lone frame 7 x 4.9 us although a level is ~1 us of work (a flag-linked multi-level kernel was measured too: ~5 us per hand-off
// through memory, 62 us).  Here ALL levels are one launch without any dependency between workgroups: a workgroup owns a
// 32 x 8 tile of the TOP level and computes the whole cone under it - per level the bounding box of (what the level above
// needs, the workgroup's share of the level itself), from the level below held in LDS; it writes every pixel of its boxes, so
// neighbouring workgroups write their overlap twice - the same bytes (a pixel is one fixed function of four source pixels,
// k_resize's arithmetic).  1.7x the arithmetic of the level launches, 1/7 of the launches.  The boxes come from the host
// (prepare(): they depend on the geometry only).
#ifdef ORBHIP_CONE_PROF
__device__ unsigned long long g_cone_ticks[24];
#define CONE_MARK(i) do { if (blockIdx.x == 77 && threadIdx.x == 0) g_cone_ticks[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define CONE_MARK(i) do { } while (0)
#endif
#define CONE_TPB 1024     /* a lone wave per SIMD issues one instruction per ~4.5 cycles: four waves per SIMD share the work of a cone */
#define CONE_MAXL 8        /* pyramid levels the cone kernel handles (its table registers are unrolled over the levels) */
#define CONE_SRC_PT 12      /* bytes of the level 0 box a thread loads (all requested at once) */
struct ConeLevel { int sw, dw, dpitch, pad; long long doff; const uint2* xtab; const int* yofs; const short* ibeta; int sh, dh; };
struct ConeArgs { ConeLevel lv[MAX_LEVELS]; int nl, spitch0, buf0, bufk; };      // buf0 / bufk: bytes of the LDS image buffers (level 0 box / larger of the others)
__global__ __launch_bounds__(CONE_TPB) void k_pyr_cone(ConeArgs A, const short* __restrict__ boxes, const uint8_t* __restrict__ img, uint8_t* __restrict__ pyr, int* __restrict__ status) {
  if (blockIdx.x == 0 && threadIdx.x == 0) status[0] = 0;          // (the frame's status word: instead of a memset launch in front)
  extern __shared__ __attribute__((aligned(16))) uint8_t s_cone[];
  const int tid = threadIdx.x, nl = A.nl;
  CONE_MARK(0);
  const short* B = boxes + (size_t)blockIdx.x * nl * 4;          // [level][x0, y0, x1, y1]
  uint8_t* s_src = s_cone;                                        // level 0 box
  uint8_t* s_buf[2] = {s_cone + A.buf0, s_cone + A.buf0 + A.bufk};
  uint8_t* s_tab = s_cone + A.buf0 + 2 * A.bufk;                  // per level: columns {s0 | s1 << 16, a0 | a1 << 16}, rows {o0, o1, b0, b1}
  // ---- one batch of loads: the level 0 box and every level's table segments (rebased to the boxes).  Every value is REQUESTED
  // before the first one is stored: a load -> LDS store loop costs a lone workgroup one global round trip per iteration.
  {
    uint2 tc[CONE_MAXL]; int ty[CONE_MAXL], tb[CONE_MAXL];
#pragma unroll
    for (int l = 1; l < CONE_MAXL; l++) {
      tc[l] = make_uint2(0u, 0u); ty[l] = 0; tb[l] = 0;
      if (l < nl) {
        const ConeLevel& L = A.lv[l];
        const int x0 = B[4 * l], y0 = B[4 * l + 1], rw = B[4 * l + 2] - x0, rh = B[4 * l + 3] - y0;      // (<= 256 each: prepare())
        if (tid < rw) tc[l] = L.xtab[x0 + tid];
        if (tid < rh) { ty[l] = L.yofs[y0 + tid]; tb[l] = *(const int*)(L.ibeta + 2 * (y0 + tid)); }
      }
    }
    const int x0 = B[0], y0 = B[1], rw = B[2] - x0, n0 = rw * (B[3] - y0);      // (<= CONE_TPB * CONE_SRC_PT bytes: prepare())
    uint8_t v[CONE_SRC_PT];
#pragma unroll
    for (int u = 0; u < CONE_SRC_PT; u++) {
      const int p = CONE_TPB * u + tid;
      const int y = p / rw, x = p - y * rw;
      v[u] = p < n0 ? img[(long long)(y0 + y) * A.spitch0 + x0 + x] : (uint8_t)0;
    }
    CONE_MARK(1);
#pragma unroll
    for (int u = 0; u < CONE_SRC_PT; u++) { const int p = CONE_TPB * u + tid; if (p < n0) s_src[p] = v[u]; }
    CONE_MARK(2);
    int toff = 0;
#pragma unroll
    for (int l = 1; l < CONE_MAXL; l++) {
      if (l < nl) {
        const ConeLevel& L = A.lv[l];
        const int lx0 = B[4 * l], ly0 = B[4 * l + 1], lrw = B[4 * l + 2] - lx0, lrh = B[4 * l + 3] - ly0;
        const int px0 = B[4 * l - 4], py0 = B[4 * l - 3], prw = B[4 * l - 2] - px0;      // the box of the level below
        uint2* cols = (uint2*)(s_tab + toff);
        int4* rows = (int4*)(s_tab + toff + 8 * lrw);
        if (tid < lrw) {
          const int sx = (int)(tc[l].x & 0xFFFF), s1 = min(sx + 1, L.sw - 1);
          cols[tid] = make_uint2((uint32_t)(sx - px0) | ((uint32_t)(s1 - px0) << 16), tc[l].y);
        }
        if (tid < lrh) {
          const int sy0 = min(max(ty[l], 0), L.sh - 1), sy1 = min(max(ty[l] + 1, 0), L.sh - 1);
          rows[tid] = make_int4((sy0 - py0) * prw, (sy1 - py0) * prw, (int)(short)(tb[l] & 0xFFFF), (int)(short)(tb[l] >> 16));
        }
        toff += 8 * lrw + 16 * lrh;
      }
    }
  }
  __syncthreads();
  CONE_MARK(3);
  // ---- level by level out of LDS
  const uint8_t* prev = s_src;
  int toff = 0;
  for (int l = 1; l < nl; l++) {
    const ConeLevel& L = A.lv[l];
    const int x0 = B[4 * l], y0 = B[4 * l + 1], rw = B[4 * l + 2] - x0, rh = B[4 * l + 3] - y0;
    const uint2* cols = (const uint2*)(s_tab + toff);
    const int4* rows = (const int4*)(s_tab + toff + 8 * rw);
    uint8_t* cur = s_buf[l & 1];
    uint8_t* dst = pyr + L.doff;
    const int rq = rw >> 2;
    for (int it = tid; it < rq * rh; it += CONE_TPB) {
      const int y = it / rq, x = (it - y * rq) * 4;
      const int4 r = rows[y];
      uint32_t out = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint2 c = cols[x + j];
        const int s0 = (int)(c.x & 0xFFFF), s1 = (int)(c.x >> 16), a0 = (int)(c.y & 0xFFFF), a1 = (int)(c.y >> 16);
        const int H0 = prev[r.x + s0] * a0 + prev[r.x + s1] * a1;
        const int H1 = prev[r.y + s0] * a0 + prev[r.y + s1] * a1;
        const int v = (((r.z * (H0 >> 4)) >> 16) + ((r.w * (H1 >> 4)) >> 16) + 2) >> 2;
        if (x0 + x + j < L.dw) out |= (uint32_t)(v & 255) << (8 * j);
      }
      *(uint32_t*)(cur + y * rw + x) = out;
      *(uint32_t*)(dst + (long long)(y0 + y) * L.dpitch + x0 + x) = out;
    }
    __syncthreads();
    CONE_MARK(3 + l);
    prev = cur;
    toff += 8 * rw + 16 * rh;
  }
}

// ---------------------------------------------------------------------------- k_fast_cells
// FAST-9 "best" of one pixel for BOTH polarities at once with packed 16-bit min/max (v_pk_min_i16), on the RAW ring values:
// P[k] = (ring[k], -ring[k]) as two i16 (ONE v_mul_i32_i24 by -65535: r * (1 - 2^16) = r + ((-r) << 16)).  A window minimum then
// holds (min ring, -max ring), the maximum over the sixteen 9-arcs (max_arcs min_arc ring, -min_arcs max_arc ring) = (B, -A), and
//   best = max over arcs of min over the arc of |v - ring| with one sign = max(B - v, v - A) = halves of (B, -A) - (v, -v).
// The sixteen arcs share their 8-windows pairwise (the structure of OpenCV's cornerScore<16>): with m8[j] = min P[j .. j+7] for the
// eight ODD j, the arcs starting at j-1 and at j are min(P[j-1], m8[j]) and min(m8[j], P[j+8]), whose maximum is
// min(m8[j], max(P[j-1], P[j+8])) - 8 + 8 + 8 window minima, 8 max, 8 min, 7 max = 47 packed operations per pixel (the first
// version differenced and negated every ring value and ran the doubling on all sixteen positions: 48 + 80).
typedef short short2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ short2_t ring_pair(int r) { return __builtin_bit_cast(short2_t, __mul24(r, -65535)); }
__device__ __forceinline__ int arc9_best_packed(const short2_t P[16], int v) {
  short2_t m2[8], m4[8], m8[8];
#pragma unroll
  for (int i = 0; i < 8; i++) m2[i] = __builtin_elementwise_min(P[2 * i + 1], P[(2 * i + 2) & 15]);     // min P[j .. j+1], j = 2i + 1
#pragma unroll
  for (int i = 0; i < 8; i++) m4[i] = __builtin_elementwise_min(m2[i], m2[(i + 1) & 7]);                // min P[j .. j+3]
#pragma unroll
  for (int i = 0; i < 8; i++) m8[i] = __builtin_elementwise_min(m4[i], m4[(i + 2) & 7]);                // min P[j .. j+7]
  short2_t best = __builtin_elementwise_min(m8[0], __builtin_elementwise_max(P[0], P[9]));
#pragma unroll
  for (int i = 1; i < 8; i++)
    best = __builtin_elementwise_max(best, __builtin_elementwise_min(m8[i], __builtin_elementwise_max(P[2 * i], P[(2 * i + 9) & 15])));
  const short2_t e = best - ring_pair(v);                   // (B - v, v - A): both within +-255
  return max((int)e.x, (int)e.y);
}

// Phase timing of k_fast_cells (tools/fast_phase_prof.py builds a separate library with -DORBHIP_FAST_PROF): every wave
// adds the 100 MHz s_memrealtime ticks it spent in each phase to g_fast_prof[phase]; off in the product build.
#ifdef ORBHIP_FAST_PROF
#define FAST_PROF_WAVES (1 << 19)
__device__ unsigned int g_fast_prof[FAST_PROF_WAVES][8];        // per wave (no contention): ticks of phases 0..4, [7] = 1 when written
#define FAST_STAMP(k) do { const unsigned long long _t = __builtin_amdgcn_s_memrealtime(); t_ph[k] = (unsigned int)(_t - t_prev); t_prev = _t; } while (0)
#define FAST_STAMP_INIT unsigned long long t_prev = __builtin_amdgcn_s_memrealtime(); unsigned int t_ph[5] = {0, 0, 0, 0, 0}
#else
#define FAST_STAMP(k) do { } while (0)
#define FAST_STAMP_INIT do { } while (0)
#endif

#define FAST_TILE_LOADS 17   // global_load_lds instructions per tile, 64 dwords each: a 64 x 64 tile (pitch 68) has 1088 dwords
// instruction JJ of the direct-to-LDS tile load (compile-time recursion: one straight-line instruction per 64 dwords)
template <int JJ>
__device__ __forceinline__ void fast_tile_load(const uint8_t* base, uint8_t* tile, int lane, int ndw, int W4, uint32_t inv, uint32_t pitch) {
  if constexpr (JJ < FAST_TILE_LOADS) {
    if (64 * JJ >= ndw) return;                                // (wave-uniform)
    const uint32_t k = (uint32_t)lane + 64u * JJ;
    const uint32_t row = __umul24(k, inv) >> 16;                // (all factors < 2^24: full-rate v_mul_u32_u24 instead of quarter-rate v_mul_lo_u32)
    const uint32_t goff = __umul24(row, pitch) + 4u * (k - __umul24(row, (uint32_t)W4));
    if ((int)k < ndw)
      // (the instruction's immediate offset would be added to BOTH the global and the LDS address: the LDS position goes
      // through M0 instead - tools/ubench/lds_direct.hip)
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(base + goff),
                                       (void __attribute__((address_space(3)))*)(tile + 256 * JJ), 4, 0, 0);
    fast_tile_load<JJ + 1>(base, tile, lane, ndw, W4, inv, pitch);
  }
}

// ONE WAVE per cell (64-thread workgroups): no cross-wave barriers, lanes = columns of the cell, rows are
// walked sequentially; the ordered (row-major) emit needs only a running wave-uniform offset.
#define FAST_WPB 1      // waves per workgroup
#ifndef FAST_CPW
#define FAST_CPW 1      // cells per wave, processed one after the other (2: 0.4858 vs 0.4827 ms - workgroup dispatch is not the limiter)
#endif
// NARROW: every cell interior of the geometry is <= 32 px wide (all KITTI / VGA levels: 30-px cells): the per-row survivor masks
// are 32-bit (one v_ffbl / v_bcnt / ds_or_b32 instead of pairs) and the pre-test's lane layout is a constant.
template <bool NARROW> struct FastMask;
template <> struct FastMask<true> {
  typedef uint32_t type;
  static __device__ __forceinline__ int popc(uint32_t m) { return __popc(m); }
  static __device__ __forceinline__ int ffs0(uint32_t m) { return __ffs((int)m) - 1; }
};
template <> struct FastMask<false> {
  typedef unsigned long long type;
  static __device__ __forceinline__ int popc(unsigned long long m) { return __popcll(m); }
  static __device__ __forceinline__ int ffs0(unsigned long long m) { return __ffsll((long long)m) - 1; }
};
template <bool NARROW>
__global__ __launch_bounds__(64 * FAST_WPB) void k_fast_cells(GeomDev G, const CellDesc* __restrict__ cells,
                                                   const uint8_t* __restrict__ img0, long long img_frame_bytes,
                                                   const uint8_t* __restrict__ pyr, int* __restrict__ cell_cnt,
                                                   uint32_t* __restrict__ cell_kps, int iniTh, int minTh, int lds_per_wave, int xcd_map) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem_all[];
  const int TP = G.tile_pitch;                        // multiple of 4
  const int plane = (G.tile_h * TP + 15) & ~15;       // multiple of 16 (the score plane is cleared with 16-byte stores)
  const int lane = threadIdx.x & 63;
  typedef typename FastMask<NARROW>::type mask_t;
  // Cell order inside a frame = plain grid order; xcd_map (batches of a multiple of 8 frames, the default since round 5) deals the
  // FRAMES to the XCDs: a frame's cells - whose 36-byte tile rows share 128-byte lines with their neighbours' - then meet in one L2
  // (FETCH_SIZE 5.2 x lower, same speed now that the kernel is VALU-bound; a remap of the CELLS, ci = (b % 8) * chunk + b / 8, was
  // 40 % slower when the kernel was latency-bound in rounds 1-2).
  int f = blockIdx.y, bxi = blockIdx.x;
  if (xcd_map) {                                              // frame f on XCD f % 8 (as k_describe; host: frame count multiple of 8)
    const uint32_t lin = blockIdx.y * gridDim.x + blockIdx.x, x = lin & 7u, i = lin >> 3;
    const uint32_t q = i / gridDim.x;
    f = (int)(x + 8u * q); bxi = (int)(i - q * gridDim.x);
  }
  const int wv = FAST_WPB == 1 ? 0 : (int)(threadIdx.x >> 6);       // (one wave per workgroup: the cell index is visibly uniform, its descriptor comes by scalar loads)
  uint8_t* smem = smem_all + (size_t)wv * lds_per_wave;
  // FAST_CPW cells per wave, one after the other on the same LDS (a wave's LDS operations execute in order, so the next cell's
  // clears cannot overtake this cell's emit reads)
  for (int rep = 0; rep < FAST_CPW; rep++) {
  const int ci = (bxi * FAST_WPB + wv) * FAST_CPW + rep;
  if (ci >= G.ncells_total) return;                   // (no workgroup-wide barrier below: waves are independent)
  uint8_t* tile = smem;                               // [tile_h][TP]
  uint8_t* score = smem + plane;                      // [tile_h][TP]
  mask_t* keep = (mask_t*)(score + plane);            // [64] NMS survivors per interior row (bit = ix)
  mask_t* k20 = keep + 64;                            // [64] survivors with score >= iniTh
  int* qcnt = (int*)(k20 + 64);                       // queue length (LDS atomic counter; 16 bytes reserved)
  unsigned short* queue = (unsigned short*)(qcnt + 4);                 // pixels that passed the pre-test
  FAST_STAMP_INIT;
  const CellDesc c = cells[ci];
  const LevelDev& L = G.lv[c.level];
  const uint8_t* src = level_ptr(G, c.level, f, img0, img_frame_bytes, pyr);
  const int tw = c.x1 - c.x0, th = c.y1 - c.y0;
  const int iw = tw - 6, ih = th - 6;
  // ---- stage the cell (incl. its 3-px apron) in LDS -------------------------------------------------
  // LDS-direct loads move ALIGNED dwords: the tile is fetched from the 4-byte boundary at or before its first pixel, so tile
  // byte (row, x) lives at LDS byte row * TP + x + bsh (TP = round_up(tw, 4) + 4 leaves room for the <= 3 extra bytes)
  const uint8_t* cell_base = src + (long long)c.y0 * L.pitch + c.x0;
  const uint32_t bsh = (uint32_t)((size_t)cell_base & 3);
  const uint8_t* base_al = cell_base - bsh;
  const uint8_t* tileb = tile + bsh;
  for (int i = lane; i < plane >> 4; i += 64) ((uint4*)score)[i] = make_uint4(0u, 0u, 0u, 0u);
  keep[lane] = 0; k20[lane] = 0;
  if (lane == 0) *qcnt = 0;
  {
    // The tile goes from global memory STRAIGHT into LDS (global_load_lds_dword: no VGPR staging, no re-alignment, no LDS
    // store instructions): a wave instruction writes 64 consecutive LDS dwords, so lane l of instruction j owns tile dword
    // k = l + 64 j = (row k / W4, column word k % W4) and reads it from row * pitch + 4 * (k % W4) - an unaligned dword when
    // the cell does not start on a 4-byte boundary, which the memory system handles.  All <= 9 instructions are in flight
    // together.  The detection window keeps a 16-px margin to the image border, so the <= 7 bytes read past the right edge of
    // a tile row are inside the frame.  (The first version staged the rows through registers: 32 loads + 16 v_alignbyte +
    // 16 LDS stores and ~170 VALU per cell; this one needs ~50.)
    const int W4 = TP >> 2, ndw = th * W4;
    const uint32_t inv = (65536u + (uint32_t)W4 - 1u) / (uint32_t)W4;          // k / W4 = (k * inv) >> 16 for k < 4096, W4 <= 17
    fast_tile_load<0>(base_al, tile, lane, ndw, W4, inv, (uint32_t)L.pitch);
    __builtin_amdgcn_s_waitcnt(0);                            // vmcnt(0): the tile has landed in LDS
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
  FAST_STAMP(0);       // tile staged
  // ---- pass A: compass pre-test; survivors are queued so that the expensive score runs on dense lanes ---------
  // A 9-arc of the 16-ring always contains two ADJACENT compass points (ring 0/4/8/12), i.e. one of {0, 8} and one of
  // {4, 12}.  Necessary for a brighter corner: max(c0, c8) > v + t and max(c4, c12) > v + t, i.e.
  // min(max(c0, c8), max(c4, c12)) - v > t; darker likewise with min / max swapped.  6 min/max + 2 sub + max + 1 compare
  // per pixel, and tighter than ">= 2 of the 4 compass points" (which also admits the opposite pairs).
  int qn = 0;
  {
    // FOUR horizontally adjacent pixels per lane, from dword LDS reads: lanes = (row, column group of 4); 8 groups x 8 rows
    // when the interior is <= 32 px wide (every KITTI / VGA level), else 16 groups x 4 rows.  With D[k] = the aligned dword at
    // tile columns 4k .. 4k+3, the interior pixels 4g .. 4g+3 (tile columns 4g+3 .. 4g+6) need
    //   v, c0, c8 : bytes 3.. of {D[g], D[g+1]} of the rows 0, +3, -3      (v_alignbyte)
    //   c12       : D[g] itself,   c4 : bytes 2.. of {D[g+1], D[g+2]}
    // - seven dword reads instead of twenty byte reads per four pixels.  The bytes are split into even / odd pixels as u16
    // pairs and the min / max network runs on v_pk_{min,max,sub}_*16: ~10 VALU per pixel instead of ~18, and one prefix sum
    // + enqueue per FOUR pixels (most lanes have nothing to enqueue: 8 % of the pixels pass).
    const bool narrow = NARROW || iw <= 32;
    const int RW = narrow ? 8 : 4;
    const int g = narrow ? (lane & 7) : (lane & 15), lr = narrow ? (lane >> 3) : (lane >> 4);
    const int ix4 = 4 * g;
    // validity of this lane's four columns (bit k = column ix4 + k is inside the interior)
    const uint32_t colmask = ix4 + 3 < iw ? 15u : (ix4 < iw ? ((1u << (iw - ix4)) - 1u) : 0u);
    const uint32_t T0 = (uint32_t)minTh * 0x00010001u;
    const uint32_t qaddr = (uint32_t)(size_t)(__attribute__((address_space(3))) int*)qcnt;      // LDS byte address of the queue counter
    const int qv = (int)((3u + bsh) >> 2), q4 = 1 + (int)((2u + bsh) >> 2);       // (wave-uniform dword offsets / byte shifts)
    const uint32_t sv = (3u + bsh) & 3u, s4 = (2u + bsh) & 3u;
    for (int iy0 = 0; iy0 < ih; iy0 += RW) {
      const int iyl = iy0 + lr;
      const int iy = min(iyl, ih - 1);                          // (clamped: rows outside are masked, their reads stay inside the tile)
      // byte offsets of the three quads inside the (shifted) LDS row: c12 at 4g + bsh, v / c0 / c8 at 4g + 3 + bsh, c4 at 4g + 6 + bsh
      const int rowoff = __mul24(iy, TP);
      const uint32_t* rc = (const uint32_t*)(tile + rowoff + 3 * TP) + g;
      const uint32_t* ru = (const uint32_t*)(tile + rowoff) + g + qv;            // ring point 8 (row - 3)
      const uint32_t* rd = (const uint32_t*)(tile + rowoff + 6 * TP) + g + qv;   // ring point 0 (row + 3)
      const uint32_t e0 = rc[0], e1 = rc[1], v0 = rc[qv], v1 = rc[qv + 1], f0 = rc[q4], f1 = rc[q4 + 1], u0 = ru[0], u1 = ru[1], b0 = rd[0], b1 = rd[1];
      const uint32_t vq = __builtin_amdgcn_alignbyte(v1, v0, sv), c8q = __builtin_amdgcn_alignbyte(u1, u0, sv), c0q = __builtin_amdgcn_alignbyte(b1, b0, sv);
      const uint32_t c12q = __builtin_amdgcn_alignbyte(e1, e0, bsh), c4q = __builtin_amdgcn_alignbyte(f1, f0, s4);
      uint32_t sg[2];                                           // per parity: bit 15 / 31 set where the pixel passes
#pragma unroll
      for (int par = 0; par < 2; par++) {                       // even pixels (bytes 0, 2), odd pixels (bytes 1, 3) as u16 pairs
        // ONE instruction per quad and parity: v_and for the even bytes, v_perm (bytes 1 and 3 to the low halves, 0x0c = constant
        // zero) for the odd ones
        auto split = [&](uint32_t q) { return par ? __builtin_amdgcn_perm(0u, q, 0x0c030c01u) : (q & 0x00FF00FFu); };
        typedef unsigned short ushort2_v __attribute__((ext_vector_type(2)));
        auto U = [](uint32_t x) { return __builtin_bit_cast(ushort2_v, x); };
        auto S = [](uint32_t x) { return __builtin_bit_cast(short2_t, x); };
        const ushort2_v v2 = U(split(vq)), a0 = U(split(c0q)), a8 = U(split(c8q)), a4 = U(split(c4q)), a12 = U(split(c12q));
        const ushort2_v hi = __builtin_elementwise_min(__builtin_elementwise_max(a0, a8), __builtin_elementwise_max(a4, a12));
        const ushort2_v lo = __builtin_elementwise_max(__builtin_elementwise_min(a0, a8), __builtin_elementwise_min(a4, a12));
        const short2_t up = S(__builtin_bit_cast(uint32_t, hi)) - S(__builtin_bit_cast(uint32_t, v2));     // (values 0..255: no overflow in 16 bits)
        const short2_t dn = S(__builtin_bit_cast(uint32_t, v2)) - S(__builtin_bit_cast(uint32_t, lo));
        const short2_t m = S(T0) - __builtin_elementwise_max(up, dn);                                       // < 0  <=>  max(..) > minTh
        sg[par] = __builtin_bit_cast(uint32_t, m);
      }
      // pixel k of the lane = byte k: pixels 0 / 2 are the sign bits 15 / 31 of the even parity, pixels 1 / 3 those of the odd one;
      // t carries them at bits 14, 15, 30, 31 (one shift + one bit-field insert), two field extracts bring them to bits 0..3
      const uint32_t t = (sg[1] & 0x80008000u) | ((sg[0] >> 1) & ~0x80008000u);
      uint32_t pm = ((t >> 14) & 3u) | ((t >> 28) & 0xCu);
      pm &= (iyl < ih) ? colmask : 0u;
      if (pm) {
        // queue slots from an LDS counter (the order of the queue is irrelevant: scores go to their pixel, survivors to row
        // masks): one ds_add_rtn by the lanes that have something, instead of a 6-step DPP prefix sum by all of them
        // (inline asm: written as atomicAdd, the compiler's atomic optimizer serialises the active lanes with a readlane /
        // writelane loop to issue ONE atomic per wave - ~8 scalar steps per lane, far more than the LDS unit's own conflict handling)
        int pos;
        asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(pos) : "v"(qaddr), "v"(__popc(pm)) : "memory");
        const uint32_t rowbits = ((uint32_t)iyl << 8) | (uint32_t)ix4;
        do {
          const int k = __ffs((int)pm) - 1;
          pm &= pm - 1;
          queue[pos++] = (unsigned short)(rowbits + (uint32_t)k);
        } while (pm);
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
  qn = __builtin_amdgcn_readfirstlane(*qcnt);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
  FAST_STAMP(1);       // pre-test + queue
  // ---- pass B: FAST-9 score at the LOW threshold; one map serves both thresholds --------
  for (int k = lane; k < qn; k += 64) {
    const int q = queue[k], iy = q >> 8, ix = q & 255;
    const uint8_t* p = tileb + __mul24(iy + 3, TP) + ix + 3;
    const int v = p[0];
    short2_t d[16];
    auto mk = [&](int r) { return ring_pair(r); };
    d[0] = mk(p[3 * TP]);         d[1] = mk(p[3 * TP + 1]);   d[2] = mk(p[2 * TP + 2]);   d[3] = mk(p[TP + 3]);
    d[4] = mk(p[3]);              d[5] = mk(p[-TP + 3]);      d[6] = mk(p[-2 * TP + 2]);  d[7] = mk(p[-3 * TP + 1]);
    d[8] = mk(p[-3 * TP]);        d[9] = mk(p[-3 * TP - 1]);  d[10] = mk(p[-2 * TP - 2]); d[11] = mk(p[-TP - 3]);
    d[12] = mk(p[-3]);            d[13] = mk(p[TP - 3]);      d[14] = mk(p[2 * TP - 2]);  d[15] = mk(p[3 * TP - 1]);
    const int best = arc9_best_packed(d, v);             // corner at t  <=>  best > t ; score = best - 1
    if (best > minTh) score[__mul24(iy + 3, TP) + ix + 3] = (uint8_t)(best - 1);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
  FAST_STAMP(2);       // 9-arc score
  // ---- 3x3 non-max suppression inside the cell (frame pixels score 0), only on queued pixels --------
  int any20 = 0;
  for (int k = lane; k < qn; k += 64) {
    const int q = queue[k], iy = q >> 8, ix = q & 255;
    const uint8_t* sp = score + __mul24(iy + 3, TP) + ix + 3;
    const int v = sp[0];
    // (all nine reads unconditional and the comparison without short-circuit: one LDS round trip per queued pixel)
    const int n0 = sp[-TP - 1], n1 = sp[-TP], n2 = sp[-TP + 1], n3 = sp[-1], n4 = sp[1], n5 = sp[TP - 1], n6 = sp[TP], n7 = sp[TP + 1];
    const int nmax = max(max(max(n0, n1), max(n2, n3)), max(max(n4, n5), max(n6, n7)));
    const bool kp = v != 0 && v > nmax;
    if (kp) {
      atomicOr(&keep[iy], (mask_t)1 << ix);
      if (v >= iniTh) { atomicOr(&k20[iy], (mask_t)1 << ix); any20 = 1; }
    }
  }
  any20 = __any(any20);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
  FAST_STAMP(3);       // NMS
  // ---- ordered emit: K20 if non-empty else K7 (src/ORBextractor.cc:812-816), row by row -----------
  const mask_t* mask = any20 ? k20 : keep;
  uint32_t* out = cell_kps + ((long long)f * G.ncells_total + ci) * G.cell_cap;
  // lane r holds row r's mask; exclusive wave scan of the row populations gives every row's base offset
  const mask_t mrow = (lane < ih) ? mask[lane] : (mask_t)0;
  const int cnt = FastMask<NARROW>::popc(mrow);
  const int incl = wave_incl_scan_i32(cnt);
  const int base = __builtin_amdgcn_readlane(incl, 63);
  // lane r writes row r's survivors itself, left to right, starting at the row's base offset: the loop runs for the
  // LARGEST row population of the cell (a handful) instead of once per non-empty row, without cross-lane traffic
  {
    mask_t m = mrow;
    int pos = incl - cnt;
    const uint32_t yv = (uint32_t)(lane + 3 + c.offy) << 12;
    const uint8_t* srow = score + __mul24(lane + 3, TP) + 3;
    while (m) {
      const int ix = FastMask<NARROW>::ffs0(m);
      m &= m - 1;
      if (pos < G.cell_cap) out[pos] = (uint32_t)(ix + 3 + c.offx) | yv | ((uint32_t)srow[ix] << 24);
      pos++;
    }
  }
  if (lane == 0) cell_cnt[(long long)f * G.ncells_total + ci] = (ih > 0 && iw > 0) ? base : 0;
  FAST_STAMP(4);       // ordered emit
#ifdef ORBHIP_FAST_PROF
  { const long long wid = (long long)f * G.ncells_total + ci;
    if (lane == 0 && wid < FAST_PROF_WAVES) { for (int k = 0; k < 5; k++) g_fast_prof[wid][k] = t_ph[k]; g_fast_prof[wid][7] = 1u; } }
#endif
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
  }   // cells of this wave
}

// ---------------------------------------------------------------------------- k_octree
// Phase timing of k_octree (tools/octree_phase_prof.py builds a scratch library with -DORBHIP_OCT_PROF): thread 0 of every
// workgroup adds the 100 MHz ticks between its stamps to g_oct_prof[level][phase]; [level][15] counts workgroups, [14] sweeps.
#ifdef ORBHIP_OCT_PROF
__device__ unsigned long long g_oct_prof[MAX_LEVELS][16];
#define OCT_STAMP(k) do { if (threadIdx.x == 0) { const unsigned long long _t = __builtin_amdgcn_s_memrealtime(); atomicAdd(&g_oct_prof[level][k], _t - t_prev); t_prev = _t; } } while (0)
#define OCT_STAMP_INIT unsigned long long t_prev = __builtin_amdgcn_s_memrealtime(); const unsigned long long t_begin = t_prev
#define OCT_STAMP_END do { if (threadIdx.x == 0) { atomicMax(&g_oct_prof[level][13], t_prev - t_begin); atomicMax(&g_oct_prof[level][12], (unsigned long long)n); } } while (0)
#define OCT_COUNT(k) do { if (threadIdx.x == 0) atomicAdd(&g_oct_prof[level][k], 1ull); } while (0)
#else
#define OCT_STAMP(k) do { } while (0)
#define OCT_STAMP_INIT do { } while (0)
#define OCT_STAMP_END do { } while (0)
#define OCT_COUNT(k) do { } while (0)
#endif
#ifndef OCT_TPB
#define OCT_TPB 256
#endif                  // threads per (frame, level) workgroup: 64 / 128 / 256 / 512 / 1024 -> 0.390 / 0.234 / 0.158 / 0.198 / 0.393 ms
// exclusive scan of a[0..n) in place by an OCT_TPB-thread block; returns the total.
template <typename T>
__device__ int block_excl_scan(T* a, int n, int* s_tmp) {
  const int tid = threadIdx.x;
  const int items = (n + OCT_TPB - 1) / OCT_TPB;
  const int beg = min(tid * items, n), end = min(beg + items, n);
  int sum = 0;
  for (int i = beg; i < end; i++) sum += (int)a[i];
  const int lane = tid & 63, w = tid >> 6;
  int v = wave_incl_scan_i32(sum);
  if (lane == 63) s_tmp[w] = v;
  __syncthreads();
  int woff = 0, total = 0;
#pragma unroll
  for (int i = 0; i < OCT_TPB / 64; i++) { int t = s_tmp[i]; if (i < w) woff += t; total += t; }
  int run = woff + v - sum;
  for (int i = beg; i < end; i++) { int t = (int)a[i]; a[i] = (T)run; run += t; }
  __syncthreads();
  return total;
}

struct __attribute__((aligned(8))) Rect16 { short ulx, uly, urx, bry; };

__device__ __forceinline__ int quad_of(const Rect16& r, int x, int y, int& mx, int& my) {
  mx = r.ulx + ((r.urx - r.ulx + 1) >> 1);     // UL.x + ceil((UR.x-UL.x)/2)   (src/ORBextractor.cc:483-484)
  my = r.uly + ((r.bry - r.uly + 1) >> 1);
  return (x < mx ? 0 : 1) + (y < my ? 0 : 2);  // n1=UL n2=UR n3=BL n4=BR        (:515-525)
}

#define OCT_U 8        // key entries per thread requested together in the sweep loops
// Per-node key counts and the four child counts of a candidate node: 16-bit fields when no level of the geometry can hold
// more than 65535 candidates, 32-bit otherwise (a node of a nearly square or noisy level can own > 65535 keys in the first
// sweeps; 16-bit fields would wrap and carry into their neighbours).
template <bool WIDE> struct OctT;
template <> struct OctT<false> {
  typedef unsigned short cnt_t; typedef uint2 cc_t;
  static __device__ __forceinline__ cc_t zero() { return make_uint2(0, 0); }
  static __device__ __forceinline__ void add(cc_t* cc, int p, int q) { atomicAdd((q & 2) ? &cc[p].y : &cc[p].x, (q & 1) ? 0x10000u : 1u); }
  static __device__ __forceinline__ void get(const cc_t& v, int c4[4]) { c4[0] = (int)(v.x & 0xFFFF); c4[1] = (int)(v.x >> 16); c4[2] = (int)(v.y & 0xFFFF); c4[3] = (int)(v.y >> 16); }
};
template <> struct OctT<true> {
  typedef unsigned int cnt_t; typedef uint4 cc_t;
  static __device__ __forceinline__ cc_t zero() { return make_uint4(0, 0, 0, 0); }
  static __device__ __forceinline__ void add(cc_t* cc, int p, int q) { atomicAdd(&cc[p].x + q, 1u); }
  static __device__ __forceinline__ void get(const cc_t& v, int c4[4]) { c4[0] = (int)v.x; c4[1] = (int)v.y; c4[2] = (int)v.z; c4[3] = (int)v.w; }
};
// Node arrays of one (frame, level) workgroup.  LDS instantiation with 16-bit counters: 44 bytes per node, so that the 442 nodes of
// nfeatures = 2000 take 19.3 kB and EIGHT workgroups share a CU's 160 kB - all 2048 workgroups of a 256-frame batch are resident at
// once (at 50 bytes per node plus a separate cell-prefix array only six fitted and the kernel ran in two rounds).  The scan
// arrays are 16-bit there (values <= 4 node_cap), the cell-prefix array of the gather phase lies over everything behind rect[0]
// (nothing else is live yet), the final-phase sort keys and the processing order share the childpos rows (dead until phase G),
// the best-key array the child-count rows (dead after the last sweep).
static size_t octree_lds_bytes(int node_cap, int max_cells_level, bool wide, bool gmem) {
  const size_t scan_b = (wide || gmem) ? 4 : 2;
  const size_t per_node = 8 * 2 + (wide ? 16 : 8) + 8 + 2 * scan_b + (wide ? 4 : 2) * 2 + 2 + 2;
  const size_t nodes = (size_t)node_cap * per_node, pref = (size_t)node_cap * 8 + (size_t)(max_cells_level + 8) * 4;
  return std::max(nodes, pref) + 16;
}
// GMEM: the node arrays live in a global scratch row of the (frame, level) workgroup instead of LDS - the fallback for per-level
// quotas whose node arrays exceed the 160 kB of LDS (about 3200 keypoints in one level, i.e. nfeatures beyond ~15000; the
// reference has no such limit).  Same code, same order of operations, slower memory.
template <bool WIDE, bool GMEM>
__device__ __forceinline__ void octree_body(const GeomDev& G, const int* __restrict__ cell_cnt,
                                            const uint32_t* __restrict__ cell_kps, uint32_t* __restrict__ keys,
                                            unsigned short* __restrict__ knode, uint32_t* __restrict__ sel,
                                            int* __restrict__ sel_cnt, int* __restrict__ nkeys_out,
                                            int* __restrict__ status, uint8_t* __restrict__ gnodes, size_t gnodes_stride, const int level, const int f) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem_lds[];
  const int tid = threadIdx.x;
#ifdef ORBHIP_OCT_LEVEL_EXPERIMENT
  if (!((G.oct_level_mask >> level) & 1)) return;
#endif
  const LevelDev& Lv = G.lv[level];
  const int NC = G.node_cap, N = Lv.quota;
  uint8_t* smem;
  if constexpr (GMEM) smem = gnodes + ((size_t)f * G.nlevels + level) * gnodes_stride; else smem = smem_lds;
  // ---- LDS carve (all offsets multiples of 8) -------------------------------------------------
  Rect16* rect[2];
  rect[0] = (Rect16*)smem;
  rect[1] = rect[0] + NC;
  typedef typename OctT<WIDE>::cnt_t cnt_t;
  typedef typename OctT<WIDE>::cc_t cc_t;
  cc_t* cc = (cc_t*)(rect[1] + NC);                   // 4 child counts of a candidate node
  uint2* childpos = (uint2*)(cc + NC);                // 4 x u16 new positions (or .x = shifted position)
  typedef typename std::conditional<(WIDE || GMEM), int, unsigned short>::type scan_t;      // scan values are <= 4 NC
  scan_t* sA = (scan_t*)(childpos + NC);
  scan_t* sB = sA + NC;
  cnt_t* cnt[2];
  cnt[0] = (cnt_t*)(sB + NC);
  cnt[1] = cnt[0] + NC;
  unsigned short* candl = (unsigned short*)(cnt[1] + NC);
  short* rankOf = (short*)(candl + NC);
  unsigned short* order = (unsigned short*)((uint8_t*)childpos + 4 * (size_t)NC);   // second half of the childpos rows (first half: sort keys)
  int* s_pref = (int*)rect[1];                        // [max_cells_level + 1], gather phase only: lies over everything behind rect[0]
  __shared__ int s_tmp[16], s_m, s_nexp, s_L, s_hist[MAX_INI], s_remap[MAX_INI], s_need_remap;

  uint32_t* K = keys + (long long)f * G.keys_per_frame + Lv.key_off;
  unsigned short* KN = knode + (long long)f * G.keys_per_frame + Lv.key_off;
  const int* ccnt = cell_cnt + (long long)f * G.ncells_total + Lv.cell_begin;
  const uint32_t* ckps = cell_kps + ((long long)f * G.ncells_total + Lv.cell_begin) * G.cell_cap;
  uint32_t* SEL = sel + ((long long)f * G.nlevels + level) * G.sel_cap;

  OCT_STAMP_INIT;
  // ---- 0. gather the level's candidates in reference order (cells row-major, pixels row-major) --
  const int ncell = Lv.ncells;
  for (int c = tid; c < ncell; c += OCT_TPB) s_pref[c] = ccnt[c];
  __syncthreads();
  const int n = block_excl_scan(s_pref, ncell, s_tmp);
  if (tid == 0) { s_pref[ncell] = n; nkeys_out[f * G.nlevels + level] = n; }
  // Two instantiations share the work: the 16-bit one (always launched) takes every level with <= 65535 candidates and
  // reports overflow; the 32-bit one is launched behind it only when the geometry allows more than 65535 candidates in a
  // level, and takes exactly those (its other workgroups leave here).
  if (WIDE ? (n <= 65535 || n > Lv.kcap) : (n > 65535 && n <= Lv.kcap)) return;
  if (n > Lv.kcap) {
    if (tid == 0) { sel_cnt[f * G.nlevels + level] = 0; atomicOr(&status[f], 1); }
    return;
  }
  const int nIni = Lv.nIni;
  for (int i = tid; i < MAX_INI; i += OCT_TPB) s_hist[i] = 0;
  __syncthreads();
  // ONE pass over the keys: fetch from the cell lists, store, initial node index and its histogram (LDS int atomics, nIni <= 64)
  for (int k0 = tid; k0 < n; k0 += OCT_TPB * OCT_U) {
    uint32_t keyv[OCT_U];
#pragma unroll
    for (int u = 0; u < OCT_U; u++) {             // all OCT_U cell-list reads of this thread in flight together
      const int k = k0 + OCT_TPB * u;
      keyv[u] = 0u;
      if (k < n) {
        int lo = 0, hi = ncell;                   // largest c with pref[c] <= k
        while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (s_pref[mid] <= k) lo = mid; else hi = mid; }
        keyv[u] = ckps[(long long)lo * G.cell_cap + (k - s_pref[lo])];
      }
    }
#pragma unroll
    for (int u = 0; u < OCT_U; u++) {
      const int k = k0 + OCT_TPB * u;
      if (k < n) {
        const uint32_t key = keyv[u];
        K[k] = key;
        int x = key & 0xFFF;
        int idx = (int)__fdiv_rn((float)x, Lv.hX);     // vpIniNodes[kp.pt.x/hX]   (src/ORBextractor.cc:569)
        idx = min(idx, nIni - 1);
        KN[k] = (unsigned short)idx;
        atomicAdd(&s_hist[idx], 1);
      }
    }
  }
  __syncthreads();                                  // (the cell-prefix array is dead from here: the node arrays it lay over come alive)
  for (int i = tid; i < NC; i += OCT_TPB) { cnt[0][i] = 0; cnt[1][i] = 0; }
  __syncthreads();
  if (tid == 0) {
    int L0 = 0;
    for (int i = 0; i < nIni; i++) {
      int c = s_hist[i];
      s_remap[i] = L0;                             // (valid where c > 0)
      if (c > 0) {
        Rect16 r; r.ulx = (short)Lv.ini_x[i]; r.uly = 0; r.urx = (short)Lv.ini_x[i + 1]; r.bry = (short)Lv.winH;
        rect[0][L0] = r; cnt[0][L0] = (cnt_t)c; L0++;
      }
    }
    s_L = L0;
    s_need_remap = L0 != nIni;                     // an empty initial node is dropped (:575-590): the later ones move up
  }
  __syncthreads();
  if (s_need_remap) {
    for (int k0 = tid; k0 < n; k0 += OCT_TPB * OCT_U) {
      int pv[OCT_U];
#pragma unroll
      for (int u = 0; u < OCT_U; u++) { const int k = k0 + OCT_TPB * u; pv[u] = k < n ? (int)KN[k] : -1; }
#pragma unroll
      for (int u = 0; u < OCT_U; u++) if (pv[u] >= 0) KN[k0 + OCT_TPB * u] = (unsigned short)s_remap[pv[u]];
    }
    __syncthreads();
  }

  int cur = 0;
  int L = s_L;
  bool final_phase = false;
  OCT_STAMP(0);        // gather + initial nodes
  OCT_COUNT(15);
  while (true) {
    OCT_COUNT(14);
    const int prev = L;
    // (round 6: offsets from ONE base, not `rect[cur]` / `cnt[cur]` - through an array of pointers indexed at run time the compiler lost the
    // address space, and every read of a node's rectangle or count in the key loops was a FLAT load into the LDS aperture, waited for alone)
    const Rect16* R = rect[0] + (size_t)cur * NC;
    const cnt_t* C = cnt[0] + (size_t)cur * NC;
    // ---- A: candidates = nodes holding > 1 key, in list order ---------------------------------
    for (int p = tid; p < L; p += OCT_TPB) { sA[p] = C[p] > 1; cc[p] = OctT<WIDE>::zero(); rankOf[p] = -1; }
    if (tid == 0) { s_m = -1; s_nexp = 0; }
    __syncthreads();
    const int ncand = block_excl_scan(sA, L, s_tmp);
    if (ncand == 0) break;                         // no split possible: |L| == prevSize  (:669)
    for (int p = tid; p < L; p += OCT_TPB) if (C[p] > 1) candl[sA[p]] = (unsigned short)p;
    OCT_STAMP(1);      // A
    // ---- B: child occupancy of every candidate -------------------------------------------------
    // (the key loops read K / KN from global memory: OCT_U entries per thread are requested together - a one-entry loop
    // pays the global latency ~20 times per sweep phase at level 0, which bounded this kernel at ~0.2 ms)
    for (int k0 = tid; k0 < n; k0 += OCT_TPB * OCT_U) {
      uint32_t keyv[OCT_U]; int pv[OCT_U];
#pragma unroll
      for (int u = 0; u < OCT_U; u++) { const int k = k0 + OCT_TPB * u; const bool in = k < n; pv[u] = in ? (int)KN[k] : -1; keyv[u] = in ? K[k] : 0u; }
#pragma unroll
      for (int u = 0; u < OCT_U; u++) {
        const int p = pv[u];
        if (p >= 0 && C[p] > 1) {
          const uint32_t key = keyv[u];
          int mx, my;
          int q = quad_of(R[p], key & 0xFFF, (key >> 12) & 0xFFF, mx, my);
          OctT<WIDE>::add(cc, p, q);
        }
      }
    }
    __syncthreads();
    OCT_STAMP(2);      // B (key loop)
    // ---- C: processing order and how many candidates get split ---------------------------------
    int m = ncand;
    if (!final_phase) {
      for (int r = tid; r < ncand; r += OCT_TPB) order[r] = candl[r];
    } else {
      // sort by (count desc, list position asc): list position asc == creation desc (canonical F9)
      if constexpr (!WIDE) {
        // rank = number of candidates with a larger packed key (count << 16 | 0xFFFF - position; keys are distinct).  The keys
        // live in the childpos rows (dead until phase G) and are read back as 16-byte broadcasts, two own keys per thread and
        // trip: ~ncand / 4 LDS reads per thread instead of 2 ncand dependent ones (the pair loop took 20 us of a level-0
        // workgroup's 83, tools/octree_phase_prof.py).
        uint32_t* skeys = (uint32_t*)childpos;
        const int npad4 = (ncand + 3) & ~3;
        for (int i = tid; i < npad4; i += OCT_TPB) {
          uint32_t kv = 0u;                                  // padding: smaller than every real key (counts are >= 2)
          if (i < ncand) { const int p = candl[i]; kv = ((uint32_t)C[p] << 16) | (0xFFFFu - (uint32_t)p); }
          skeys[i] = kv;
        }
        __syncthreads();
        for (int i0 = tid; i0 < ncand; i0 += 2 * OCT_TPB) {
          const int i1 = i0 + OCT_TPB;
          const uint32_t my0 = skeys[i0], my1 = i1 < ncand ? skeys[i1] : 0xFFFFFFFFu;
          int r0 = 0, r1 = 0;
          for (int j = 0; j < npad4; j += 4) {
            const uint4 k4 = *(const uint4*)(skeys + j);
            r0 += (int)(k4.x > my0) + (int)(k4.y > my0) + (int)(k4.z > my0) + (int)(k4.w > my0);
            r1 += (int)(k4.x > my1) + (int)(k4.y > my1) + (int)(k4.z > my1) + (int)(k4.w > my1);
          }
          order[r0] = (unsigned short)(0xFFFFu - (my0 & 0xFFFFu));
          if (i1 < ncand) order[r1] = (unsigned short)(0xFFFFu - (my1 & 0xFFFFu));
        }
      } else {
        for (int i = tid; i < ncand; i += OCT_TPB) {
          int p = candl[i], cp = (int)C[p], r = 0;
          for (int j = 0; j < ncand; j++) { int pj = candl[j], cj = (int)C[pj]; r += (cj > cp) || (cj == cp && pj < p); }
          order[r] = (unsigned short)p;
        }
      }
      __syncthreads();
      for (int r = tid; r < ncand; r += OCT_TPB) {
        int c4[4]; OctT<WIDE>::get(cc[order[r]], c4);
        sA[r] = (c4[0] != 0) + (c4[1] != 0) + (c4[2] != 0) + (c4[3] != 0) - 1;
        sB[r] = sA[r];
      }
      __syncthreads();
      block_excl_scan(sA, ncand, s_tmp);
      for (int r = tid; r < ncand; r += OCT_TPB) {
        int before = L + sA[r], after = before + sB[r];
        if (after >= N && before < N) s_m = r + 1;       // first split that reaches |L| >= N  (:730-731)
      }
      __syncthreads();
      if (s_m >= 0) m = s_m;
    }
    __syncthreads();
    OCT_STAMP(final_phase ? 4 : 3);      // C (plain / final-phase sort)
    // ---- D/E/F: ranks, creation bases (rank order), split prefix (list order) -------------------
    for (int r = tid; r < m; r += OCT_TPB) {
      int p = order[r];
      rankOf[p] = (short)r;
      int c4[4]; OctT<WIDE>::get(cc[p], c4);
      sA[r] = (c4[0] != 0) + (c4[1] != 0) + (c4[2] != 0) + (c4[3] != 0);
    }
    __syncthreads();
    const int TC = block_excl_scan(sA, m, s_tmp);          // sA[r] = creation index of r's first child
    for (int p = tid; p < L; p += OCT_TPB) sB[p] = rankOf[p] >= 0;
    __syncthreads();
    block_excl_scan(sB, L, s_tmp);                         // sB[p] = #split nodes before p
    // ---- G: new list = reverse(children in creation order) ++ (old list minus split nodes) -------
    Rect16* Rn = rect[0] + (size_t)(cur ^ 1) * NC;
    cnt_t* Cn = cnt[0] + (size_t)(cur ^ 1) * NC;
    int nexp = 0;
    for (int p = tid; p < L; p += OCT_TPB) {
      int r = rankOf[p];
      if (r >= 0) {
        Rect16 rc = R[p];
        int mx, my; quad_of(rc, 0, 0, mx, my);
        int c4[4]; OctT<WIDE>::get(cc[p], c4);
        int e = sA[r];
        unsigned short pos4[4] = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 4; q++) {
          if (c4[q] > 0) {
            int pos = TC - 1 - e; e++;
            Rect16 ch;
            ch.ulx = (q & 1) ? (short)mx : rc.ulx; ch.urx = (q & 1) ? rc.urx : (short)mx;
            ch.uly = (q & 2) ? (short)my : rc.uly; ch.bry = (q & 2) ? rc.bry : (short)my;
            Rn[pos] = ch; Cn[pos] = (cnt_t)c4[q];
            pos4[q] = (unsigned short)pos;
            nexp += c4[q] > 1;
          }
        }
        childpos[p] = make_uint2(pos4[0] | ((uint32_t)pos4[1] << 16), pos4[2] | ((uint32_t)pos4[3] << 16));
      } else {
        int pos = TC + p - sB[p];
        Rn[pos] = R[p]; Cn[pos] = C[p];
        childpos[p] = make_uint2((uint32_t)pos, 0);
      }
    }
    if (nexp) atomicAdd(&s_nexp, nexp);
    __syncthreads();
    OCT_STAMP(5);      // D-G
    // ---- H: move the keys -----------------------------------------------------------------------
    for (int k0 = tid; k0 < n; k0 += OCT_TPB * OCT_U) {
      uint32_t keyv[OCT_U]; int pv[OCT_U];
#pragma unroll
      for (int u = 0; u < OCT_U; u++) { const int k = k0 + OCT_TPB * u; const bool in = k < n; pv[u] = in ? (int)KN[k] : -1; keyv[u] = in ? K[k] : 0u; }
#pragma unroll
      for (int u = 0; u < OCT_U; u++) {
        const int p = pv[u];
        if (p < 0) continue;
        const int k = k0 + OCT_TPB * u;
        uint2 cp = childpos[p];
        if (rankOf[p] >= 0) {
          const uint32_t key = keyv[u];
          int mx, my;
          int q = quad_of(R[p], key & 0xFFF, (key >> 12) & 0xFFF, mx, my);
          uint32_t wv = (q & 2) ? cp.y : cp.x;
          KN[k] = (unsigned short)((q & 1) ? (wv >> 16) : (wv & 0xFFFF));
        } else {
          KN[k] = (unsigned short)cp.x;
        }
      }
    }
    L = TC + L - m;
    const int nToExpand = s_nexp;
    cur ^= 1;
    __syncthreads();
    OCT_STAMP(6);      // H (key loop)
    if (L >= N || L == prev) break;                              // (:669-672, :734-735)
    if (!final_phase && L + 3 * nToExpand > N) final_phase = true;   // (:673)
  }
  __syncthreads();
  // ---- best key of every node: max response, earliest candidate on ties (:741-760) ---------------
  unsigned int* best = (unsigned int*)cc;                  // (the child-count rows are dead after the last sweep)
  for (int p = tid; p < L; p += OCT_TPB) best[p] = 0;
  __syncthreads();
  for (int k0 = tid; k0 < n; k0 += OCT_TPB * OCT_U) {
    uint32_t keyv[OCT_U]; int pv[OCT_U];
#pragma unroll
    for (int u = 0; u < OCT_U; u++) { const int k = min(k0 + OCT_TPB * u, n - 1); pv[u] = (int)KN[k]; keyv[u] = K[k]; }      // (no branch around the loads: the compiler requested them pair by pair, each pair waited for before its atomic)
    asm volatile("" : "+v"(pv[0]), "+v"(pv[1]), "+v"(pv[2]), "+v"(pv[3]), "+v"(pv[4]), "+v"(pv[5]), "+v"(pv[6]), "+v"(pv[7]),
                      "+v"(keyv[0]), "+v"(keyv[1]), "+v"(keyv[2]), "+v"(keyv[3]), "+v"(keyv[4]), "+v"(keyv[5]), "+v"(keyv[6]), "+v"(keyv[7]));
#pragma unroll
    for (int u = 0; u < OCT_U; u++) if (k0 + OCT_TPB * u < n) atomicMax(&best[pv[u]], ((keyv[u] >> 24) << 24) | (0xFFFFFFu - (unsigned)(k0 + OCT_TPB * u)));
  }
  __syncthreads();
  for (int p = tid; p < L; p += OCT_TPB) {
    if (p < G.sel_cap) SEL[p] = K[0xFFFFFFu - (best[p] & 0xFFFFFFu)];
  }
  if (tid == 0) {
    sel_cnt[f * G.nlevels + level] = L;
    if (L > G.sel_cap) atomicOr(&status[f], 2);
  }
  OCT_STAMP(7);        // best key per node + output
  OCT_STAMP_END;
}
// (frame-major order, levels fastest.  Tried: level-major, all level-0 workgroups - the longest - dispatched first: 0.163 ->
// 0.190 ms; the 256 long workgroups then compete with each other for the same CUs' LDS pipes instead of being interleaved
// with short ones.)
template <bool WIDE, bool GMEM>
__global__ __launch_bounds__(OCT_TPB) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_octree(GeomDev G, const int* __restrict__ cell_cnt,
                                                const uint32_t* __restrict__ cell_kps, uint32_t* __restrict__ keys,
                                                unsigned short* __restrict__ knode, uint32_t* __restrict__ sel,
                                                int* __restrict__ sel_cnt, int* __restrict__ nkeys_out,
                                                int* __restrict__ status, uint8_t* __restrict__ gnodes, size_t gnodes_stride) {
  octree_body<WIDE, GMEM>(G, cell_cnt, cell_kps, keys, knode, sel, sel_cnt, nkeys_out, status, gnodes, gnodes_stride, (int)blockIdx.x, (int)blockIdx.y);
}
// A lone frame whose geometry allows more than 65535 candidates in a level needs both instantiations (the wide one leaves at once
// unless a level really has that many): one launch of 2 x nlevels workgroups instead of two launches - 4.7 us of launch floor per
// frame on the latency path.
__global__ __launch_bounds__(OCT_TPB) void k_octree_pair(GeomDev G, const int* __restrict__ cell_cnt, const uint32_t* __restrict__ cell_kps,
                                                         uint32_t* __restrict__ keys, unsigned short* __restrict__ knode, uint32_t* __restrict__ sel,
                                                         int* __restrict__ sel_cnt, int* __restrict__ nkeys_out, int* __restrict__ status) {
  const int bx = (int)blockIdx.x, f = (int)blockIdx.y;
  if (bx < G.nlevels) octree_body<false, false>(G, cell_cnt, cell_kps, keys, knode, sel, sel_cnt, nkeys_out, status, nullptr, 0, bx, f);
  else octree_body<true, false>(G, cell_cnt, cell_kps, keys, knode, sel, sel_cnt, nkeys_out, status, nullptr, 0, bx - G.nlevels, f);
}

// ---------------------------------------------------------------------------- k_blur7
#define BLUR_TW 128
#define BLUR_TH 64
#if !defined(BLUR_COL_SLIDE) || !defined(ORBHIP_EXPERIMENTS)
#undef BLUR_COL_SLIDE
#define BLUR_COL_SLIDE 0      // 1: a thread owns eight CONSECUTIVE output rows (14 LDS reads + 56 unpacks per thread instead of 56 + 224, bit-exact) - 0.407 ms against 0.392
#endif
__device__ __forceinline__ int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) { if (i < 0) i = -i; else i = 2 * (n - 1) - i; }
  return i;
}
// 128 x 32 output tile per workgroup.  Row pass straight from global memory (each thread: 4 outputs
// from 10 source bytes = 3 aligned dwords when the row pitch allows) into a u16 LDS plane
// (255*257 fits 16 bits); column pass reads 7 x ds_read_b64 per 4 outputs and stores one dword.
// VAR: which OpenCV fixed-point Gaussian the taps restate (orbx_set_opencv_variant): 0 = cvRound(g * 256) = {18,34,49,55,49,34,18}
// (sum 257: OpenCV <= 3.4.1, i.e. the 2.4.11 / 3.2 the reference names), 1 = the error-diffused 8.8 taps {18,34,48,56,48,34,18} (sum 256)
// of the "bit-exact" ufixedpoint16 path of later versions.  Same two passes, same final (sum + 2^15) >> 16.
template <int VAR>
__global__ __launch_bounds__(256) void k_blur7(GeomDev G, const BlurTile* __restrict__ tiles,
                                               const uint8_t* __restrict__ img0, long long img_frame_bytes,
                                               const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur) {
  __shared__ __attribute__((aligned(16))) unsigned short s_mid[BLUR_TH + 6][BLUR_TW];
  const BlurTile t = tiles[blockIdx.x];
  const int f = blockIdx.y, tid = threadIdx.x;
  const LevelDev& L = G.lv[t.level];
  const uint8_t* src = level_ptr(G, t.level, f, img0, img_frame_bytes, pyr);
  uint8_t* dst = blur + (long long)f * G.blur_frame_bytes + L.blur_off;
  const int x0 = t.tx * BLUR_TW, y0 = t.ty * BLUR_TH;
  const int cg = tid & 31, rr = tid >> 5;            // column group (4 px) / row within a pass of 8 rows
  const int xo = x0 + 4 * cg;                        // first output column of this thread
  const bool aligned = ((L.pitch & 3) == 0) && (((size_t)src & 3) == 0);
  const bool inner = (xo >= 4) && (xo + 8 <= L.w);   // bytes xo-4 .. xo+7 all inside the row
  // Row pass.  (Tried again in round 2, with the kernel 100 % VALU-busy: two v_dot4_u32_u8 per output on v_alignbyte windows of
  // the 12-byte block - 14 instructions per four outputs instead of ~40, bit-exact - runs 0.450 ms against 0.392: the dot
  // instructions cost more issue time than the SDWA byte-select multiply-adds they replace.)
  // (Round 4: the row sums fit 16 bits exactly - 255 x 257 = 65535 - so the interior was rewritten on PACKED u16: nine v_perm byte pairs,
  // three v_pk_add_u16 + four v_pk_mul / v_pk_mad_u16 per output pair, 23 instructions per four outputs, stored as they come; bit-exact,
  // and slower: 1.01 ms against 0.92 beside FAST + octree, 137.2 k against 140.8 k frames/s pipelined on the same box.)
  for (int yy = rr; yy < BLUR_TH + 6; yy += 8) {
    const int sy = reflect101(y0 + yy - 3, L.h);
    const uint8_t* row = src + (uint32_t)sy * (uint32_t)L.pitch;      // (32-bit unsigned row offset on the wave-uniform level base: no per-lane 64-bit multiply)
    int b[10];                                         // source bytes xo-3 .. xo+6
    if (xo < L.w) {
      if (inner && aligned) {
        const uint32_t w0 = *(const uint32_t*)(row + xo - 4), w1 = *(const uint32_t*)(row + xo), w2 = *(const uint32_t*)(row + xo + 4);
        b[0] = (w0 >> 8) & 255; b[1] = (w0 >> 16) & 255; b[2] = w0 >> 24;
        b[3] = w1 & 255; b[4] = (w1 >> 8) & 255; b[5] = (w1 >> 16) & 255; b[6] = w1 >> 24;
        b[7] = w2 & 255; b[8] = (w2 >> 8) & 255; b[9] = (w2 >> 16) & 255;
      } else if (inner) {
#pragma unroll
        for (int k = 0; k < 10; k++) b[k] = row[xo - 3 + k];
      } else {
#pragma unroll
        for (int k = 0; k < 10; k++) b[k] = row[reflect101(xo - 3 + k, L.w)];
      }
      // taps cvRound(g*256) = {18,34,49,55,49,34,18}
      unsigned short o[4];
#pragma unroll
      for (int j = 0; j < 4; j++)
        o[j] = (unsigned short)(18 * (b[j] + b[j + 6]) + 34 * (b[j + 1] + b[j + 5]) + (VAR ? 48 : 49) * (b[j + 2] + b[j + 4]) + (VAR ? 56 : 55) * b[j + 3]);
      *(uint2*)&s_mid[yy][4 * cg] = make_uint2((uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16));
    }
  }
  __syncthreads();
  if (xo >= L.w) return;
#if BLUR_COL_SLIDE == 0
  {
    uint8_t* drow = dst + (uint32_t)(y0 + rr) * (uint32_t)L.bpitch + (uint32_t)xo;
    const uint32_t dstep = 8u * (uint32_t)L.bpitch;
    for (int yy = rr; yy < BLUR_TH; yy += 8, drow += dstep) {
      const int y = y0 + yy;
      if (y >= L.h) break;
      int acc[4] = {0, 0, 0, 0};
      const int taps[7] = {18, 34, VAR ? 48 : 49, VAR ? 56 : 55, VAR ? 48 : 49, 34, 18};
#pragma unroll
      for (int k = 0; k < 7; k++) {
        const uint2 m = *(const uint2*)&s_mid[yy + k][4 * cg];
        acc[0] += taps[k] * (int)(m.x & 0xFFFF); acc[1] += taps[k] * (int)(m.x >> 16);
        acc[2] += taps[k] * (int)(m.y & 0xFFFF); acc[3] += taps[k] * (int)(m.y >> 16);
      }
      uint32_t out = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) { int v = (acc[j] + (1 << 15)) >> 16; v = v > 255 ? 255 : v; out |= (uint32_t)v << (8 * j); }
      *(uint32_t*)drow = out;                                      // bpitch is a multiple of 64 >= w
    }
  }
#else
#include "experiments/blur_col_slide.inc"
#endif
}

// ---------------------------------------------------------------------------- k_blur7_mfma (round 5)
// The same 7 x 7 Gaussian on the MATRIX cores (VERDICT r4 next #5a): the extract kernels bind on the VALU issue slots (k_blur7: 43
// lane-instructions per pixel, 29 % of the front-end's VALU work) while the matrix pipes idle.  A separable filter is two banded
// matrix products, and every product here is exact in integers, so the result is bit for bit k_blur7's:
//   row pass     Mid = Src x Tr      v_mfma_i32_16x16x64_i8: A = 16 rows x 64 source bytes (a lane: 16 consecutive bytes of one row,
//                                    pixel - 128 as int8), B = Toeplitz band of the taps (int8, <= 56), C starts at 128 x sum(taps):
//                                    the accumulators ARE the 16-bit row sums of k_blur7
//   column pass  Out^T = Mid^T x Tc  Mid split into high and low bytes (each - 128 as int8): two products per 16 x 16 outputs,
//                                    Out = (256 H + L + (256 x 128 + 128) x sum(taps) + 2^15) >> 16, clamped to 255
// One wave blurs 58 rows x 48 columns from 64 rows x 64 columns of source with NO LDS and no barrier: the row-to-lane and
// column-to-lane assignments of the products are chosen so that every result lands where its next use wants it -
//   * row-pass product b (of 4) takes the source rows R0 + 16 (m >> 2) + 4 b + (m & 3) as its 16 M-rows: lane (i, g) then holds in
//     acc[b][r] the row sums of source rows R0 + 16 g + 4 b + r of ONE column - sixteen consecutive rows = its 16 K-bytes of the
//     column pass, in K order;
//   * row-pass column block cb (of 3) produces the output columns c + 12 (n >> 2) + 4 cb + (n & 3): in the column pass (A = Mid^T: M =
//     the block's columns) lane (i, g) receives rows m = 4 g + r = columns c + 12 g + 4 cb + r of output row 16 s + i, so the three
//     blocks give it TWELVE CONSECUTIVE output bytes of one row: one 12-byte store.
// The index k inside an instruction pairs byte j of lane group g of A with byte j of lane group g of B (csrc/orb_matcher.hip), so
// "k = 16 g + j" below is a convention the Toeplitz tables (host-built, g_blur_toep) share with the source operand.
// VALU per wave: ~16 per source operand, 6 per four row sums (pack + sign), 11 per four outputs: ~7 lane-instructions per pixel.
#define BM_TW 192                   /* output columns per workgroup (4 waves x 48) */
#define BM_TH 58                    /* output rows per chunk */
#define BM_RC 4                     /* chunks (of 58 rows) a wave walks down its 48 columns: the Toeplitz operands are loaded once, the next
                                       chunk's source is in flight during the products (one-chunk waves were dispatch- and latency-bound:
                                       0.425 ms per 256 frames at 26 % VALU-busy) */
typedef int bm_v4i __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(1))) bm_u128 { uint32_t x, y, z, w; };     // a 16-byte load with no alignment promise
typedef uint32_t bm_u32x3 __attribute__((ext_vector_type(3), aligned(4)));        // 12 bytes, 4-byte aligned: one global_store_dwordx3
__device__ bm_v4i g_blur_toep[2][7][64];   // [variant][3 row-pass blocks | 4 column-pass blocks][lane]: the lane's 16 K-bytes of the Toeplitz operand
template <int VAR>
__global__ __launch_bounds__(256) void k_blur7_mfma(GeomDev G, const BlurTile* __restrict__ tiles,
                                                    const uint8_t* __restrict__ img0, long long img_frame_bytes,
                                                    const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur) {
  const BlurTile t = tiles[blockIdx.x];                 // (level, tx = 192-column strip, ty = first chunk, pad = chunks)
  const int f = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const LevelDev& L = G.lv[t.level];
  const int c = t.tx * BM_TW + 48 * wv;                 // first output column of this wave
  if (c >= L.w) return;
  const uint8_t* src = level_ptr(G, t.level, f, img0, img_frame_bytes, pyr);
  uint8_t* dst = blur + (long long)f * G.blur_frame_bytes + L.blur_off;
  const int li = lane & 15, g = lane >> 4;
  const int cb0 = c - 4 + 16 * g;                       // this lane's 16 source bytes: columns cb0 .. cb0 + 15 (k = 16 g + j)
  const int rsub = 16 * (li >> 2) + (li & 3);           // product b, M-row li = source row R0 + rsub + 4 b
  constexpr int SUM = VAR ? 256 : 257;
  bm_v4i Tr[3], Tc[4];
#pragma unroll
  for (int q = 0; q < 3; q++) Tr[q] = g_blur_toep[VAR][q][lane];
#pragma unroll
  for (int q = 0; q < 4; q++) Tc[q] = g_blur_toep[VAR][3 + q][lane];
  const bool cols_inside = c - 4 >= 0 && c + 60 <= L.w;               // wave-uniform: every lane's 16 bytes lie inside the row
  // (bytes beyond the columns a tap reaches multiply zeros, but must be readable: the slow path reflects them too)
  auto load_src = [&](int R0, uint32_t (&v)[4][4]) {
    const bool rows_inside = R0 >= 0 && R0 + 64 <= L.h;               // wave-uniform
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const int ry = R0 + rsub + 4 * b;
      const int sy = rows_inside ? ry : reflect101(ry, L.h);
      const uint8_t* row = src + (uint32_t)sy * (uint32_t)L.pitch;
      if (cols_inside) {
        const bm_u128 u = *(const bm_u128*)(row + cb0);
        v[b][0] = u.x; v[b][1] = u.y; v[b][2] = u.z; v[b][3] = u.w;
      } else {
#pragma unroll
        for (int q = 0; q < 4; q++) {
          v[b][q] = 0;
#pragma unroll
          for (int j = 0; j < 4; j++) v[b][q] |= (uint32_t)row[reflect101(cb0 + 4 * q + j, L.w)] << (8 * j);
        }
      }
    }
  };
  uint32_t cur[4][4], nxt[4][4];
  load_src(t.ty * BM_TH - 3, cur);
  for (int ch = 0; ch < t.pad; ch++) {
    const int R0 = (t.ty + ch) * BM_TH - 3;             // first source row of the chunk's 64
    if (ch + 1 < t.pad) load_src(R0 + BM_TH, nxt);
    // ---- row pass: the row sums of this lane's column of block cb, rows R0 + 16 g .. + 15, split into sign-flipped high / low bytes
    bm_v4i A[4];
#pragma unroll
    for (int b = 0; b < 4; b++)
      A[b] = (bm_v4i){(int)(cur[b][0] ^ 0x80808080u), (int)(cur[b][1] ^ 0x80808080u), (int)(cur[b][2] ^ 0x80808080u), (int)(cur[b][3] ^ 0x80808080u)};
    bm_v4i Bhi[3], Blo[3];
#pragma unroll
    for (int cb = 0; cb < 3; cb++) {
      int hi[4], lo[4];
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const bm_v4i acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[b], Tr[cb], (bm_v4i){128 * SUM, 128 * SUM, 128 * SUM, 128 * SUM}, 0, 0, 0);
        const uint32_t p01 = (uint32_t)acc[0] | ((uint32_t)acc[1] << 16), p23 = (uint32_t)acc[2] | ((uint32_t)acc[3] << 16);     // (sums <= 65535)
        lo[b] = (int)(__builtin_amdgcn_perm(p23, p01, 0x06040200u) ^ 0x80808080u);
        hi[b] = (int)(__builtin_amdgcn_perm(p23, p01, 0x07050301u) ^ 0x80808080u);
      }
      Blo[cb] = (bm_v4i){lo[0], lo[1], lo[2], lo[3]};
      Bhi[cb] = (bm_v4i){hi[0], hi[1], hi[2], hi[3]};
    }
    // ---- column pass + store: output row R0 + 3 + 16 s + li, columns c + 12 g .. + 11
    constexpr int K1 = (256 * 128 + 128) * SUM + (1 << 15);
    const int ocol = c + 12 * g;
    const bool whole = ocol + 12 <= L.bpitch;           // (bpitch and ocol are multiples of 4)
#pragma unroll
    for (int s = 0; s < 4; s++) {
      uint32_t o[3];
#pragma unroll
      for (int cb = 0; cb < 3; cb++) {
        const bm_v4i H = __builtin_amdgcn_mfma_i32_16x16x64_i8(Bhi[cb], Tc[s], (bm_v4i){0, 0, 0, 0}, 0, 0, 0);
        const bm_v4i Lq = __builtin_amdgcn_mfma_i32_16x16x64_i8(Blo[cb], Tc[s], (bm_v4i){K1, K1, K1, K1}, 0, 0, 0);
        uint32_t tv[4];
#pragma unroll
        for (int r = 0; r < 4; r++) { const uint32_t x = ((uint32_t)H[r] << 8) + (uint32_t)Lq[r]; tv[r] = x < 0x00FFFFFFu ? x : 0x00FFFFFFu; }     // byte 2 = min(value >> 16, 255)
        o[cb] = __builtin_amdgcn_perm(tv[1], tv[0], 0x0c0c0602u) | __builtin_amdgcn_perm(tv[3], tv[2], 0x06020c0cu);
      }
      const int orow = R0 + 3 + 16 * s + li;
      if (16 * s + li < BM_TH && orow < L.h) {
        uint8_t* d = dst + (uint32_t)orow * (uint32_t)L.bpitch + (uint32_t)ocol;
        if (whole) *(bm_u32x3*)d = (bm_u32x3){o[0], o[1], o[2]};      // (lanes g = 0..3 of a row: 48 contiguous bytes)
        else {
#pragma unroll
          for (int q = 0; q < 3; q++) if (ocol + 4 * q + 4 <= L.bpitch) *(uint32_t*)(d + 4 * q) = o[q];
        }
      }
    }
#pragma unroll
    for (int b = 0; b < 4; b++)
#pragma unroll
      for (int q = 0; q < 4; q++) cur[b][q] = nxt[b][q];
  }
}

// ---------------------------------------------------------------------------- k_resize_mfma (round 5)
// cv::resize INTER_LINEAR with the HORIZONTAL pass on the matrix cores (k_resize above is the VALU kernel and the fallback
// for geometries this one does not cover).  H[sy][dx] = S[sy][sx] a0 + S[sy][sx + 1] a1 is a product of the source rows with a
// two-banded weight matrix; the weights (0 .. 2048) go in as two int8 digits, a = 32 ah + al, the pixels as int8 (pixel - 128):
//   H^T = Wh^T S'^T * 32 + Wl^T S'^T + 128 (a0 + a1)        (exact in integers: H is k_resize's H bit for bit)
// as v_mfma_i32_16x16x64_i8 with A = the weight digits (M = 16 output columns, K = 64 source columns; host tables, one set per
// 48-column chunk of a level) and B = the source (K = 64 source bytes of N = 16 consecutive source rows; a lane: 16 consecutive
// bytes of one row).  C[m][n]: lane (n, g) holds output columns 4 g .. 4 g + 3 of block cb for SOURCE ROW n - and the block's
// columns are assigned c + 12 (m >> 2) + 4 cb + (m & 3), so the three blocks give a lane twelve consecutive output bytes.
// The vertical pass, v = (((b0 (H0 >> 4)) >> 16) + ((b1 (H1 >> 4)) >> 16) + 2) >> 2, truncates per term and stays on the VALU:
// source row sy0 = the lane's own row, sy1 = sy0 + 1 = the NEXT LANE's (one DPP row shift; the row index is clamped at load time,
// so the bottom edge's sy1 = sy0 comes out by itself); a source row is sy0 of at most one output row (scale > 1), which the row
// table names.  Lane 15 of a block has no next lane: blocks advance by 15 source rows.  A wave walks RM_RB blocks down its 48
// columns with the next block's source in flight.  ~10 VALU instructions per output pixel instead of ~40.
#define RM_RB 4                      /* 15-row blocks per wave */
struct RmLevel { const uint8_t* tab; size_t oW, oC, oC0, oRow; int nchunks, nblocks, sw, sh, dw, dh; };
struct alignas(8) RmRow { int32_t dy; uint32_t b01; };       // output row whose yofs is this source row (-1: none), b0 | b1 << 16
__global__ __launch_bounds__(256) void k_resize_mfma(RmLevel R, const uint8_t* __restrict__ src, int spitch, long long sframe, uint8_t* __restrict__ dst, int dpitch,
                                                     long long dframe, int ntx) {
  const int f = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int tx = blockIdx.x % ntx, tb = blockIdx.x / ntx;                 // 192-column strip, first block of RM_RB
  const int t = tx * 4 + wv;                                              // 48-column chunk of this wave
  if (t >= R.nchunks) return;
  const int li = lane & 15, g = lane >> 4;
  const uint8_t* S = src + (long long)f * sframe;
  uint8_t* D = dst + (long long)f * dframe;
  const int C0 = ((const int*)(R.tab + R.oC0))[t];
  bm_v4i Wh[3], Wl[3], Ci[3];
  {
    const bm_v4i* W = (const bm_v4i*)(R.tab + R.oW) + (size_t)t * 6 * 64;
    const bm_v4i* Cc = (const bm_v4i*)(R.tab + R.oC) + (size_t)t * 3 * 4;
#pragma unroll
    for (int cb = 0; cb < 3; cb++) { Wh[cb] = W[(2 * cb) * 64 + lane]; Wl[cb] = W[(2 * cb + 1) * 64 + lane]; Ci[cb] = Cc[cb * 4 + g]; }
  }
  const RmRow* rows = (const RmRow*)(R.tab + R.oRow);
  const int cb0 = C0 + 16 * g;                                            // this lane's 16 source bytes
  const bool cols_inside = C0 >= 0 && C0 + 64 <= R.sw;                    // wave-uniform
  auto load_src = [&](int R0, uint32_t (&v)[4]) {
    const int sy = min(max(R0 + li, 0), R.sh - 1);
    const uint8_t* row = S + (long long)sy * spitch;
    if (cols_inside) {
      const bm_u128 u = *(const bm_u128*)(row + cb0);
      v[0] = u.x; v[1] = u.y; v[2] = u.z; v[3] = u.w;
    } else {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        v[q] = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) v[q] |= (uint32_t)row[min(max(cb0 + 4 * q + j, 0), R.sw - 1)] << (8 * j);      // (columns no weight reaches: any readable byte)
      }
    }
  };
  const int b_lo = tb * RM_RB, b_hi = min(b_lo + RM_RB, R.nblocks);
  uint32_t cur[4], nxt[4] = {0, 0, 0, 0};
  load_src(15 * b_lo, cur);
  const int ocol = 48 * t + 12 * g;
  const bool whole = ocol + 12 <= dpitch;
  for (int b = b_lo; b < b_hi; b++) {
    const int R0 = 15 * b;
    if (b + 1 < b_hi) load_src(R0 + 15, nxt);
    const RmRow ri = rows[R0 + li];                                       // (the table is padded to 15 nblocks + 1 entries)
    const bm_v4i Bs = {(int)(cur[0] ^ 0x80808080u), (int)(cur[1] ^ 0x80808080u), (int)(cur[2] ^ 0x80808080u), (int)(cur[3] ^ 0x80808080u)};
    const uint32_t b0 = ri.b01 & 0xFFFFu, b1 = ri.b01 >> 16;
    uint32_t o[3];
#pragma unroll
    for (int cb = 0; cb < 3; cb++) {
      const bm_v4i Ph = __builtin_amdgcn_mfma_i32_16x16x64_i8(Wh[cb], Bs, (bm_v4i){0, 0, 0, 0}, 0, 0, 0);
      const bm_v4i Pl = __builtin_amdgcn_mfma_i32_16x16x64_i8(Wl[cb], Bs, Ci[cb], 0, 0, 0);
      uint32_t px[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int H0 = (Ph[r] << 5) + Pl[r];                              // this lane's source row
        const int H1 = __builtin_amdgcn_update_dpp(0, H0, 0x101, 0xF, 0xF, false);      // row_shl:1 - the next lane's (next source row's) H
        const uint32_t v = (((b0 * (uint32_t)(H0 >> 4)) >> 16) + ((b1 * (uint32_t)(H1 >> 4)) >> 16) + 2u) >> 2;
        px[r] = v & 255u;
      }
      o[cb] = px[0] | (px[1] << 8) | (px[2] << 16) | (px[3] << 24);
    }
    if (li < 15 && ri.dy >= 0) {
      uint8_t* d = D + (long long)ri.dy * dpitch + ocol;
      if (whole) *(bm_u32x3*)d = (bm_u32x3){o[0], o[1], o[2]};
      else {
#pragma unroll
        for (int q = 0; q < 3; q++) if (ocol + 4 * q + 4 <= dpitch) *(uint32_t*)(d + 4 * q) = o[q];
      }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) cur[q] = nxt[q];
  }
}

// ---------------------------------------------------------------------------- k_describe
__constant__ __attribute__((aligned(16))) signed char c_pattern[1024];
__constant__ int c_umax[16];

// cv::fastAtan2 (degrees), scalar OpenCV 2.4/3.x form, un-contracted
__device__ __forceinline__ float fast_atan2_deg(float y, float x, float p1, float p3, float p5, float p7) {
  const float eps = 2.2204460492503131e-16f;
  float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = __fdiv_rn(ay, __fadd_rn(ax, eps));
    c2 = __fmul_rn(c, c);
    a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
  } else {
    c = __fdiv_rn(ax, __fadd_rn(ay, eps));
    c2 = __fmul_rn(c, c);
    a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
  }
  if (x < 0) a = __fsub_rn(180.f, a);
  if (y < 0) a = __fsub_rn(360.f, a);
  return a;
}

// canonical deterministic sin/cos in double (identical operation order to oracle det_sincos)
__device__ __forceinline__ void det_sincos(double x, double* s_out, double* c_out) {
  const double TWO_OVER_PI = 6.36619772367581382433e-01;
  const double PIO2_HI = 1.57079632673412561417e+00;
  const double PIO2_LO = 6.07710050650619224932e-11;
  double kd = rint(__dmul_rn(x, TWO_OVER_PI));
  int k = (int)kd;
  double r = __dsub_rn(__dsub_rn(x, __dmul_rn(kd, PIO2_HI)), __dmul_rn(kd, PIO2_LO));
  double z = __dmul_rn(r, r);
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  double ps = __dadd_rn(__dmul_rn(__dadd_rn(__dmul_rn(__dadd_rn(__dmul_rn(__dadd_rn(__dmul_rn(__dadd_rn(__dmul_rn(S6, z), S5), z), S4), z), S3), z), S2), z), S1);
  double sn = __dadd_rn(r, __dmul_rn(__dmul_rn(r, z), ps));
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double pc = __dadd_rn(__dmul_rn(__dadd_rn(__dmul_rn(__dadd_rn(__dmul_rn(__dadd_rn(__dmul_rn(__dadd_rn(__dmul_rn(C6, z), C5), z), C4), z), C3), z), C2), z), C1);
  double cs = __dadd_rn(__dsub_rn(1.0, __dmul_rn(0.5, z)), __dmul_rn(__dmul_rn(z, z), pc));
  switch (k & 3) {
    case 0: *s_out = sn; *c_out = cs; break;
    case 1: *s_out = cs; *c_out = -sn; break;
    case 2: *s_out = -sn; *c_out = -cs; break;
    default: *s_out = -cs; *c_out = sn; break;
  }
}

// The blurred 37 x 37 reach of a keypoint's rotated BRIEF pattern (|x|, |y| <= 13 -> radius 18) goes straight into LDS as 37
// rows of 10 aligned dwords (LDS-direct loads, 64 dwords per instruction), requested as soon as the keypoint is known - i.e.
// together with the orientation's own loads - so that the 512 byte gathers of the descriptor become LDS reads instead of a
// second dependent global round trip.
#define DESC_PATCH_R 18
#define DESC_PATCH_W4 10
#define DESC_PATCH_DW (37 * DESC_PATCH_W4)      // 370 dwords, 6 instructions
#define DESC_PATCH_LDS 384
template <int JJ>
__device__ __forceinline__ void desc_patch_load(const uint8_t* base_al, uint32_t* lds, int lane, uint32_t pitch) {
  if constexpr (JJ < 6) {
    const uint32_t k = (uint32_t)lane + 64u * JJ;
    const uint32_t row = (k * 6554u) >> 16;                    // k / 10 for k < 16384
    const uint32_t goff = row * pitch + 4u * (k - row * (uint32_t)DESC_PATCH_W4);
    if (k < (uint32_t)DESC_PATCH_DW)
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(base_al + goff),
                                       (void __attribute__((address_space(3)))*)(lds + 64 * JJ), 4, 0, 0);
    desc_patch_load<JJ + 1>(base_al, lds, lane, pitch);
  }
}

#define DESC_WPB 4      // keypoints (= waves) per workgroup (1 / 2 / 4 / 8 / 16: 0.575 / 0.548 / 0.530 / 0.551 / 0.587 ms)
__global__ __launch_bounds__(64 * DESC_WPB) void k_describe(GeomDev G, const uint32_t* __restrict__ sel,
                                                  const int* __restrict__ sel_cnt, const int* __restrict__ status,
                                                  const uint8_t* __restrict__ img0, long long img_frame_bytes,
                                                  const uint8_t* __restrict__ pyr, const uint8_t* __restrict__ blur,
                                                  orbx_keypoint* __restrict__ kps, uint8_t* __restrict__ desc,
                                                  int cap, int* __restrict__ counts, float p1, float p3, float p5,
                                                  float p7, float factorPI, int xcd_map) {
  // pattern pair k as FLOATS (x0, x1, y0, y1): one 16-byte LDS read per pair, and the two x and the two y of a pair sit in
  // consecutive registers for the packed-f32 rotation below (the byte form cost a sign-extension + conversion per coordinate)
  __shared__ __attribute__((aligned(16))) float s_patf[256][4];
  // intensity-centroid weights as byte vectors: for patch row |v| and source dword i (bytes k = 4i .. 4i+3 of the 31-byte row,
  // u = k - 15), s_icm holds [|u| <= umax[|v|]] and s_ick holds k * [..] - the row sums become v_dot4_u32_u8 on whole dwords
  __shared__ __attribute__((aligned(16))) uint32_t s_icm[16][8], s_ick[16][8];
  __shared__ __attribute__((aligned(16))) uint32_t s_patch[DESC_WPB][2][DESC_PATCH_LDS];
  for (int q = threadIdx.x; q < 256; q += 64 * DESC_WPB) {
    { const signed char* pc = c_pattern + 4 * q;
      s_patf[q][0] = (float)pc[0]; s_patf[q][1] = (float)pc[2]; s_patf[q][2] = (float)pc[1]; s_patf[q][3] = (float)pc[3]; }
    const int which = q >> 7, rv = (q >> 3) & 15, i4 = q & 7, d = c_umax[rv];
    uint32_t w = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int k = 4 * i4 + j, u = k - 15;
      const bool in = k < 31 && u >= -d && u <= d;
      w |= (uint32_t)(in ? (which ? k : 1) : 0) << (8 * j);
    }
    if (which) s_ick[rv][i4] = w; else s_icm[rv][i4] = w;
  }
  __syncthreads();
  // XCD-aware frame placement: workgroups are dealt round-robin to the 8 XCDs by their LINEAR id, and every XCD has its own
  // L2.  With the plain (block, frame) order the ~900 workgroups of one frame are spread over all eight L2s and every patch row
  // is fetched from the fabric up to 8 times (PMC: 4.65 MB per frame for 2.9 MB of level data).  xcd_map puts frame f on XCD
  // f % 8: linear id L -> XCD x = L % 8, position i = L / 8 inside the XCD's queue -> frame x + 8 (i / blocks), block i % blocks.
  int f = blockIdx.y, bx = blockIdx.x;
  if (xcd_map) {                                              // (host: only when the frame count is a multiple of 8)
    const uint32_t lin = blockIdx.y * gridDim.x + blockIdx.x, x = lin & 7u, i = lin >> 3;
    const uint32_t q = i / gridDim.x;
    f = (int)(x + 8u * q); bx = (int)(i - q * gridDim.x);
  }
  const int lane = threadIdx.x & 63;
  const int half = lane >> 5, hl = lane & 31;
  // The grid is laid out per LEVEL (every level gets capacity / (2 DESC_WPB) workgroups): level and position inside the level
  // come from the block index, so the keys are requested together with the level counts and the frame status.
  // TWO keypoints per wave, one per 32-lane half: the intensity-centroid sums need 31 lanes per keypoint, the 256 BRIEF tests
  // are 8 per lane, and every lane computes the orientation of its own half - the per-wave instruction stream barely grows
  // while the global requests in flight per wave double (the kernel is bound by its chain of dependent global round trips:
  // 67 % VALU-busy at 6.3 waves per SIMD with one keypoint per wave).
  int level = 0;
#pragma unroll
  for (int l = 1; l < MAX_LEVELS; l++) if (l < G.nlevels && bx >= G.lv[l].dblk_begin) level = l;
  const LevelDev& L = G.lv[level];
  const int pos = ((bx - L.dblk_begin) * DESC_WPB + (threadIdx.x >> 6)) * 2 + half;
  const int st_f = status[f];
  // level offsets (levels concatenated 0..L-1, src/ORBextractor.cc:1075-1104): one lane per level
  const int cl = (lane < G.nlevels) ? min(sel_cnt[f * G.nlevels + lane], G.sel_cap) : 0;
  const uint32_t key_any = sel[((long long)f * G.nlevels + level) * G.sel_cap + min(pos, G.sel_cap - 1)];   // (speculative: position may be past the level's count)
  const int incl = wave_incl_scan_i32(cl);
  const int total = __builtin_amdgcn_readlane(incl, 63);                     // (lanes >= nlevels hold 0)
  const int cl_level = __builtin_amdgcn_readlane(cl, level);
  const int i = __builtin_amdgcn_readlane(incl, level) - cl_level + pos;     // keypoint index inside the frame
  const bool bad = st_f != 0 || total > cap;
  if (bx == 0 && threadIdx.x == 0) counts[f] = bad ? (st_f != 0 ? -1 : -2) : total;
  const bool valid = pos < cl_level;
  if (bad || !__any(valid)) return;
  // an empty half repeats the other half's keypoint (its loads stay inside the image, nothing is stored)
  const uint32_t key = valid ? key_any : (uint32_t)__builtin_amdgcn_readlane((int)key_any, 0);
  const int cx = (int)(key & 0xFFF) + L.minBX, cy = (int)((key >> 12) & 0xFFF) + L.minBY;
  const int resp = (int)(key >> 24);
  // ---- the blurred patches of both keypoints of the wave are requested first (they are needed last) ----
  const uint8_t* bimg = blur + (long long)f * G.blur_frame_bytes + L.blur_off + (long long)cy * L.bpitch + cx;
  uint32_t bsh;                                               // byte offset of this half's patch inside its first LDS dword
  {
    const int wv = (int)(threadIdx.x >> 6);
    const int cx0 = __builtin_amdgcn_readlane(cx, 0), cy0 = __builtin_amdgcn_readlane(cy, 0), cx1 = __builtin_amdgcn_readlane(cx, 32), cy1 = __builtin_amdgcn_readlane(cy, 32);
    const uint8_t* lev = blur + (long long)f * G.blur_frame_bytes + L.blur_off;
    const uint8_t* p0 = lev + (long long)(cy0 - DESC_PATCH_R) * L.bpitch + (cx0 - DESC_PATCH_R);
    const uint8_t* p1 = lev + (long long)(cy1 - DESC_PATCH_R) * L.bpitch + (cx1 - DESC_PATCH_R);
    const uint32_t s0 = (uint32_t)((size_t)p0 & 3), s1 = (uint32_t)((size_t)p1 & 3);
    desc_patch_load<0>(p0 - s0, &s_patch[wv][0][0], lane, (uint32_t)L.bpitch);
    desc_patch_load<0>(p1 - s1, &s_patch[wv][1][0], lane, (uint32_t)L.bpitch);
    bsh = half ? s1 : s0;
  }
  // ---- IC_Angle on the un-blurred level (src/ORBextractor.cc:77-104): lanes 0..30 of each half = rows -15..15 ------
  const uint8_t* img = level_ptr(G, level, f, img0, img_frame_bytes, pyr);
  int m10 = 0, m01 = 0;
  if (hl < 31) {
    // the row's 31 bytes as NINE aligned dwords re-cut with v_alignbyte (the 19-px border keeps the <= 3 bytes before and
    // the <= 5 after inside the image row) and two v_dot4_u32_u8 per dword against the weight vectors: 9 requests and ~30
    // VALU per row instead of 31 byte requests and 155 VALU - the kernel is bound by the request rate of its gathers
    const int v = hl - 15, rv = v < 0 ? -v : v;
    const uint8_t* rp = img + (long long)(cy + v) * L.pitch + (cx - 15);
    const uint32_t sh = (uint32_t)((size_t)rp & 3);
    const uint32_t* ap = (const uint32_t*)(rp - sh);
    uint32_t dw[9];
#pragma unroll
    for (int q = 0; q < 9; q++) dw[q] = ap[q];
    const uint4 mA = *(const uint4*)&s_icm[rv][0], mB = *(const uint4*)&s_icm[rv][4];
    const uint4 kA = *(const uint4*)&s_ick[rv][0], kB = *(const uint4*)&s_ick[rv][4];
    const uint32_t mw[8] = {mA.x, mA.y, mA.z, mA.w, mB.x, mB.y, mB.z, mB.w}, kw[8] = {kA.x, kA.y, kA.z, kA.w, kB.x, kB.y, kB.z, kB.w};
    uint32_t rs = 0, ks = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const uint32_t w = __builtin_amdgcn_alignbyte(dw[q + 1], dw[q], sh);       // source bytes k = 4q .. 4q+3
      rs = __builtin_amdgcn_udot4(w, mw[q], rs, false);
      ks = __builtin_amdgcn_udot4(w, kw[q], ks, false);
    }
    m10 = (int)ks - 15 * (int)rs;                                                // sum of (k - 15) * I over the circle's row
    m01 = v * (int)rs;
  }
  // sums over each half: rows of 16 lanes, then row 0 -> 1 and row 2 -> 3; lane 31 / 63 hold the half totals
  m10 += dpp_i32<0xB1, 0xF>(m10); m01 += dpp_i32<0xB1, 0xF>(m01);
  m10 += dpp_i32<0x4E, 0xF>(m10); m01 += dpp_i32<0x4E, 0xF>(m01);
  m10 += dpp_i32<0x141, 0xF>(m10); m01 += dpp_i32<0x141, 0xF>(m01);
  m10 += dpp_i32<0x140, 0xF>(m10); m01 += dpp_i32<0x140, 0xF>(m01);
  m10 += dpp_i32<0x142, 0xA>(m10); m01 += dpp_i32<0x142, 0xA>(m01);
  {
    const int a10 = __builtin_amdgcn_readlane(m10, 31), b10 = __builtin_amdgcn_readlane(m10, 63);
    const int a01 = __builtin_amdgcn_readlane(m01, 31), b01 = __builtin_amdgcn_readlane(m01, 63);
    m10 = half ? b10 : a10; m01 = half ? b01 : a01;
  }
  const float angle = fast_atan2_deg((float)m01, (float)m10, p1, p3, p5, p7);
  // ---- rotated BRIEF-256 on the blurred level (src/ORBextractor.cc:107-147) ---------------------
  const float ang_rad = __fmul_rn(angle, factorPI);
  double sd, cd;
  det_sincos((double)ang_rad, &sd, &cd);
  const float a = (float)cd, b = (float)sd;
  (void)bimg;
  __builtin_amdgcn_s_waitcnt(0);                              // the patches have landed (their requests precede the orientation's loads)
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
  const uint8_t* lpatch = (const uint8_t*)&s_patch[threadIdx.x >> 6][half][0] + bsh + DESC_PATCH_R * (4 * DESC_PATCH_W4) + DESC_PATCH_R;   // pattern origin
  // pair index pr = 32 j + hl: consecutive lanes read consecutive pattern words, and the wave ballot of test j holds descriptor
  // dword j of the first keypoint in its low half and of the second keypoint in its high half (bit pr & 7 of byte pr >> 3)
  // Both points of a pair at once on packed f32 (v_pk_mul_f32 / v_pk_add_f32, the same IEEE operations in the same order as the
  // scalar form, no contraction): FY = X b + Y a, FX = X a - Y b with X = (x0, x1), Y = (y0, y1).  cvRound = round-half-even is
  // ONE more packed add of 1.5 * 2^23: the integer then sits in the low mantissa bits (0x4B400000 + i for |i| < 2^22), the
  // multiply-add of the LDS address uses only the low 24 bits of the row term (0x400000 + iy) and the constant parts go into
  // the patch base.  8 packed operations per pair instead of 8 conversions + 12 multiplies / adds + 8 roundings.
  typedef float float2_v __attribute__((ext_vector_type(2)));
  const float2_v A2 = {a, a}, B2 = {b, b}, MAGIC = {12582912.0f, 12582912.0f};
  const uint32_t lbase = (uint32_t)(size_t)(__attribute__((address_space(3))) const uint8_t*)lpatch
                         - (0x400000u * (uint32_t)(4 * DESC_PATCH_W4) + 0x4B400000u);
  uint32_t mydw = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const float4 pf = *(const float4*)&s_patf[32 * j + hl][0];
    const float2_v X = {pf.x, pf.y}, Y = {pf.z, pf.w};
    const float2_v FY = X * B2 + Y * A2;
    const float2_v FX = X * A2 - Y * B2;
    const float2_v RY = FY + MAGIC, RX = FX + MAGIC;
    int t[2];
#pragma unroll
    for (int s = 0; s < 2; s++) {
      const uint32_t iyb = __builtin_bit_cast(uint32_t, s ? RY.y : RY.x), ixb = __builtin_bit_cast(uint32_t, s ? RX.y : RX.x);
      const uint32_t addr = __umul24(iyb, (uint32_t)(4 * DESC_PATCH_W4)) + ixb + lbase;
      t[s] = *(const __attribute__((address_space(3))) uint8_t*)(size_t)addr;
    }
    const unsigned long long bal = __ballot(t[0] < t[1]);
    const uint32_t mine = half ? (uint32_t)(bal >> 32) : (uint32_t)bal;
    if (hl == j) mydw = mine;
  }
  if (valid && hl < 8) *(uint32_t*)(desc + ((long long)f * cap + i) * 32 + 4 * hl) = mydw;
  if (valid && hl == 0) {
    orbx_keypoint kp;
    kp.x = (float)cx; kp.y = (float)cy;
    if (level != 0) { kp.x = __fmul_rn(kp.x, L.scale); kp.y = __fmul_rn(kp.y, L.scale); }   // (:1095-1101)
    kp.size = L.patch; kp.angle = angle; kp.response = (float)resp; kp.octave = level; kp.class_id = -1;
    kps[(long long)f * cap + i] = kp;
  }
}

// ============================================================================ host side
inline int cv_round(double v) { return (int)std::nearbyint(v); }

}  // namespace orbhip

using namespace orbhip;

static std::atomic<unsigned long long> g_ctx_generation{0};
struct orbx_ctx {
  const unsigned long long generation = ++g_ctx_generation;      // a resident frame remembers (pointer, generation): a context re-created at the same address is another producer
  int nfeatures, nlevels, iniTh, minTh, device;
  double scaleFactor;
  std::vector<float> scale, inv_scale, sigma2, inv_sigma2;
  std::vector<int> quota, umax;
  float atan_p[4], factorPI;
  // shape-dependent state
  int w = 0, h = 0, stride = 0, nframes = 0;
  GeomDev G;
  std::vector<CellDesc> cells;
  std::vector<BlurTile> btiles, mtiles, mtiles1;      // k_blur7's 128 x 64 tiles, k_blur7_mfma's strips of BM_RC chunks (batches) / of one chunk (a lone frame: latency)
  DevBuf d_cells, d_btiles, d_mtiles, d_mtiles1, d_tab;      // tables
  int blur_mfma = 1;                  // k_blur7_mfma (default) / k_blur7 (ORBHIP_BLUR_MFMA=0: the VALU kernel, for A/B runs)
  std::vector<size_t> tab_xofs, tab_ialpha, tab_yofs, tab_ibeta;   // byte offsets into d_tab per level
  struct RmHost { size_t oW = 0, oC = 0, oC0 = 0, oRow = 0; int nchunks = 0, nblocks = 0; bool ok = false; };
  std::vector<RmHost> rm;              // k_resize_mfma's tables per level (ok = false: the level takes k_resize)
  int resize_mfma = 1;                 // ORBHIP_RESIZE_MFMA=0: every level on the VALU kernel k_resize (A/B runs)
  size_t tab_cone = 0; int cone_wgs = 0, cone_buf0 = 0, cone_bufk = 0; size_t cone_lds = 0;      // k_pyr_cone: boxes in d_tab, grid, LDS layout (cone_wgs == 0: not available)
  DevBuf d_pyr, d_blur, d_cellcnt, d_cellkps, d_keys, d_knode, d_sel, d_selcnt, d_nkeys, d_status, d_octnodes;
  DevBuf d_img, d_out;                     // host-API staging: image; {counts | keypoints | descriptors} in one block
  void* h_pin = nullptr; size_t h_bytes = 0;   // pinned host mirror of both
  size_t fast_lds = 0, octree_lds = 0, octree_lds_wide = 0;
  bool octree_wide = false;           // some level can hold > 65535 candidates: 32-bit node counters (k_octree<true, .>)
  int fast_xcd = 1;                   // k_fast_cells with frame f on XCD f % 8 (batches of a multiple of 8 frames).  Round 5: default - with the kernel
                                      // VALU-bound the mapping costs nothing any more (176.3 k against 175.2 k frames/s) and its 36-byte tile rows
                                      // meet their 128-byte lines in ONE L2: FETCH_SIZE 860 -> 164 MB per 256-frame launch (ORBHIP_FAST_XCD=0 in
                                      // an experiments build restores the plain order; rounds 1-2 measured it 40 % slower, latency-bound then)
  bool fast_narrow = false;           // k_fast_cells<true>: all cell interiors <= 32 px wide
  int desc_xcd = 1;                   // k_describe: frame f on XCD f % 8 (ORBHIP_DESC_XCD=0 restores the plain order)
  bool octree_gmem = false;           // node arrays larger than the LDS: global scratch rows (k_octree<., true>)
  size_t octree_row = 0;
  // last call (for introspection)
  const uint8_t* last_img0 = nullptr; long long last_img_frame_bytes = 0; int last_nframes = 0;
  bool const_uploaded = false;
  // optional per-stage HIP-event timing (bench.py roofline): 6 events per batch call
  bool profiling = false;
  std::vector<hipEvent_t> prof_events, side_events;
  // the blur pass only depends on the pyramid: it runs on a side stream concurrently with FAST + octree
  hipStream_t side = nullptr;        // batches: normal priority
  hipStream_t side_hi = nullptr;     // a lone frame inside the Tracking chain: greatest priority (created on first use)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int lone_side_mode = 0;      // side-stream mode of a LONE frame for the next run_batch call (set by orbx_extract_chained, reset by run_batch)
  int blur_variant = 0;        // orbx_set_opencv_variant: which OpenCV GaussianBlur the taps restate (k_blur7)
  int overlap_blur = -1;       // -1: by batch size (see run_batch); 0: one stream; 1: k_blur7 on the side stream beside FAST + octree; 2: beside the octree only
};

static int build_tables(orbx_ctx* c) {
  const int nl = c->nlevels;
  c->scale.resize(nl); c->sigma2.resize(nl); c->inv_scale.resize(nl); c->inv_sigma2.resize(nl);
  c->scale[0] = 1.0f; c->sigma2[0] = 1.0f;
  for (int i = 1; i < nl; i++) {
    c->scale[i] = (float)(c->scale[i - 1] * c->scaleFactor);        // src/ORBextractor.cc:421
    c->sigma2[i] = c->scale[i] * c->scale[i];
  }
  for (int i = 0; i < nl; i++) { c->inv_scale[i] = 1.0f / c->scale[i]; c->inv_sigma2[i] = 1.0f / c->sigma2[i]; }
  c->quota.resize(nl);
  float factor = (float)(1.0f / c->scaleFactor);
  float nDesired = c->nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nl));
  int sum = 0;
  for (int l = 0; l < nl - 1; l++) { c->quota[l] = cv_round(nDesired); sum += c->quota[l]; nDesired *= factor; }
  c->quota[nl - 1] = std::max(c->nfeatures - sum, 0);
  c->umax.assign(HALF_PATCH + 1, 0);
  int v, v0, vmax = (int)std::floor(HALF_PATCH * std::sqrt(2.f) / 2 + 1);
  int vmin = (int)std::ceil(HALF_PATCH * std::sqrt(2.f) / 2);
  const double hp2 = HALF_PATCH * HALF_PATCH;
  for (v = 0; v <= vmax; ++v) c->umax[v] = cv_round(std::sqrt(hp2 - v * v));
  for (v = HALF_PATCH, v0 = 0; v >= vmin; --v) {
    while (c->umax[v0] == c->umax[v0 + 1]) ++v0;
    c->umax[v] = v0;
    ++v0;
  }
  const float k = (float)(180.0 / 3.14159265358979323846);
  c->atan_p[0] = 0.9997878412794807f * k; c->atan_p[1] = -0.3258083974640975f * k;
  c->atan_p[2] = 0.1555786518463281f * k; c->atan_p[3] = -0.04432655554792128f * k;
  c->factorPI = (float)(3.14159265358979323846 / 180.f);
  return 0;
}

// (re)build geometry + workspace for a batch of nframes images of w x h (row stride `stride`)
static int prepare(orbx_ctx* c, int w, int h, int stride, int nframes) {
  const bool same_shape = (c->w == w && c->h == h && c->stride == stride);
  if (same_shape && nframes <= c->nframes) return 0;
  ORBHIP_REQUIRE(w >= 2 * EDGE_THRESHOLD + 8 && h >= 2 * EDGE_THRESHOLD + 8, ORBHIP_EINVAL, "image too small");
  ORBHIP_REQUIRE(w <= 4095 && h <= 4095, ORBHIP_EINVAL, "image larger than 4095 px per side");
  const int nl = c->nlevels;
  GeomDev& G = c->G;
  if (!same_shape) {
    std::memset(&G, 0, sizeof(G));
    G.nlevels = nl;
    c->cells.clear(); c->btiles.clear(); c->mtiles.clear(); c->mtiles1.clear();
    long long pyr_off = 0, blur_off = 0;
    int key_off = 0, tile_w = 8, tile_h = 8, cell_cap = 1, max_cells = 1, node_cap = MAX_INI + 8, sel_cap = 8, desc_blocks = 0;
    std::vector<uint8_t> tab;
    std::vector<std::vector<int>> h_xofs, h_yofs;                 // (host copies for the cone boxes below)
    c->tab_xofs.assign(nl, 0); c->tab_ialpha.assign(nl, 0); c->tab_yofs.assign(nl, 0); c->tab_ibeta.assign(nl, 0);
    for (int l = 0; l < nl; l++) {
      LevelDev& L = G.lv[l];
      float s = c->inv_scale[l];
      L.w = cv_round((float)w * s); L.h = cv_round((float)h * s);      // src/ORBextractor.cc:1112
      ORBHIP_REQUIRE(L.w >= 1 && L.h >= 1, ORBHIP_EINVAL, "image too small for the requested number of pyramid levels");
      L.pitch = (l == 0) ? stride : round_up(L.w, 64);
      L.bpitch = round_up(L.w, 64);
      L.pyr_off = pyr_off; if (l > 0) pyr_off += (long long)L.pitch * L.h;
      L.blur_off = blur_off; blur_off += (long long)L.bpitch * L.h;
      L.scale = c->scale[l];
      L.patch = (float)(int)(PATCH_SIZE * c->scale[l]);                 // :837
      L.quota = c->quota[l];
      // detection window and cell grid (:773-787)
      const int minBX = EDGE_THRESHOLD - 3, minBY = minBX;
      const int maxBX = L.w - EDGE_THRESHOLD + 3, maxBY = L.h - EDGE_THRESHOLD + 3;
      L.minBX = minBX; L.minBY = minBY; L.winW = maxBX - minBX; L.winH = maxBY - minBY;
      const float W = 30;
      const float width = (float)(maxBX - minBX), height = (float)(maxBY - minBY);
      const int nCols = (int)(width / W), nRows = (int)(height / W);
      L.cell_begin = (int)c->cells.size();
      if (nCols >= 1 && nRows >= 1) {
        const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
        for (int i = 0; i < nRows; i++) {
          const float iniY = (float)(minBY + i * hCell);
          float maxY = iniY + hCell + 6;
          if (iniY >= maxBY - 3) continue;
          if (maxY > maxBY) maxY = (float)maxBY;
          for (int j = 0; j < nCols; j++) {
            const float iniX = (float)(minBX + j * wCell);
            float maxX = iniX + wCell + 6;
            if (iniX >= maxBX - 6) continue;
            if (maxX > maxBX) maxX = (float)maxBX;
            CellDesc cd;
            cd.level = (short)l; cd.x0 = (short)iniX; cd.y0 = (short)iniY; cd.x1 = (short)maxX; cd.y1 = (short)maxY;
            cd.offx = (short)(j * wCell); cd.offy = (short)(i * hCell); cd.pad = 0;
            c->cells.push_back(cd);
            int tw = cd.x1 - cd.x0, th = cd.y1 - cd.y0;
            tile_w = std::max(tile_w, tw); tile_h = std::max(tile_h, th);
            int iw = std::max(tw - 6, 0), ih = std::max(th - 6, 0);
            cell_cap = std::max(cell_cap, ((iw + 1) / 2) * ((ih + 1) / 2));
          }
        }
      }
      L.ncells = (int)c->cells.size() - L.cell_begin;
      max_cells = std::max(max_cells, L.ncells);
      // octree initial nodes (:543-563)
      // (levels too small to hold a cell produce no candidates; the reference divides by zero there)
      int nIni = (L.ncells > 0) ? (int)std::round(static_cast<float>(L.winW) / L.winH) : 1;
      if (nIni < 1) nIni = 1;
      ORBHIP_REQUIRE(nIni <= MAX_INI, ORBHIP_EINVAL, "aspect ratio too extreme (more than 64 initial octree nodes)");
      L.nIni = nIni;
      L.hX = (L.ncells > 0) ? static_cast<float>(L.winW) / nIni : 1.0f;
      for (int i = 0; i <= nIni; i++) L.ini_x[i] = (int)(L.hX * static_cast<float>(i));
      node_cap = std::max(node_cap, std::max(L.quota + 8, 4 * nIni + 8));
      sel_cap = std::max(sel_cap, std::max(L.quota + 4, 4 * nIni + 4));   // the first octree sweep can return 4 * nIni > N nodes
      L.dblk_begin = desc_blocks; L.dblk_count = (std::max(L.quota + 4, 4 * nIni + 4) + 2 * DESC_WPB - 1) / (2 * DESC_WPB); desc_blocks += L.dblk_count;
      long long theo = (long long)L.ncells * cell_cap;
      int keycap_max = KEYCAP_MAX;
      if (const char* e = std::getenv("ORBHIP_KEYCAP")) keycap_max = std::max(64, std::min(KEYCAP_MAX, atoi(e)));   // test hook for the overflow path
      L.kcap = (int)std::min<long long>(std::max<long long>(theo, 64), keycap_max);
      L.key_off = key_off; key_off += round_up(L.kcap, 4);
      // resize tables: level l from level l-1
      if (l > 0) {
        const int sw = G.lv[l - 1].w, sh = G.lv[l - 1].h, dw = L.w, dh = L.h;
        double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
        double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
        std::vector<int> xofs(dw), yofs(dh);
        std::vector<short> ia(2 * dw), ib(2 * dh);
        h_xofs.resize(nl); h_yofs.resize(nl);
        auto sat = [](int v) { return (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); };
        for (int dx = 0; dx < dw; dx++) {
          float fx = (float)((dx + 0.5) * scale_x - 0.5);
          int sx = (int)std::floor(fx);
          fx -= sx;
          if (sx < 0) { fx = 0; sx = 0; }
          if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
          xofs[dx] = sx;
          ia[2 * dx] = sat(cv_round((1.f - fx) * 2048)); ia[2 * dx + 1] = sat(cv_round(fx * 2048));
        }
        for (int dy = 0; dy < dh; dy++) {
          float fy = (float)((dy + 0.5) * scale_y - 0.5);
          int sy = (int)std::floor(fy);
          fy -= sy;
          yofs[dy] = sy;
          ib[2 * dy] = sat(cv_round((1.f - fy) * 2048)); ib[2 * dy + 1] = sat(cv_round(fy * 2048));
        }
        auto push = [&](const void* p, size_t bytes) {
          size_t off = (tab.size() + 15) / 16 * 16;
          tab.resize(off + bytes);
          std::memcpy(tab.data() + off, p, bytes);
          return off;
        };
        const int dw4 = round_up(dw, 4);                      // padded with copies of the last column: a thread reads its 4 entries as two 16-byte loads
        std::vector<uint32_t> xt(2 * (size_t)dw4);
        for (int dx4 = 0; dx4 < dw4; dx4++) {
          const int dx = std::min(dx4, dw - 1);
          xt[2 * dx4] = (uint32_t)(xofs[dx] & 0xFFFF) | ((uint32_t)(uint16_t)ia[2 * dx] << 16);
          xt[2 * dx4 + 1] = (uint32_t)(uint16_t)ia[2 * dx] | ((uint32_t)(uint16_t)ia[2 * dx + 1] << 16);   // both weights, v_dot2 operand order
        }
        c->tab_xofs[l] = push(xt.data(), xt.size() * 4);
        c->tab_ialpha[l] = 0;
        c->tab_yofs[l] = push(yofs.data(), yofs.size() * 4);
        c->tab_ibeta[l] = push(ib.data(), ib.size() * 2);
        h_xofs[l] = xofs; h_yofs[l] = yofs;
        // ---- k_resize_mfma's tables (see the kernel): per 48-column chunk the weight digits as MFMA A operands + the accumulator
        // start values, per source row the output row it is sy0 of.  Conditions, checked here: a chunk's source span fits 64 columns
        // (true for scale factors up to ~1.3), yofs strictly increasing and non-negative (true for every downscale).
        {
          if (c->rm.size() != (size_t)nl) c->rm.assign(nl, orbx_ctx::RmHost());
          orbx_ctx::RmHost& M = c->rm[l];
          M = orbx_ctx::RmHost();
          const int nchunks = (dw + 47) / 48;
          bool ok = true;
          std::vector<int> c0(nchunks);
          for (int t = 0; t < nchunks && ok; t++) {
            const int lo = xofs[48 * t], hi = std::min(xofs[std::min(48 * t + 47, dw - 1)] + 1, sw - 1);
            c0[t] = lo & ~3;
            if (hi - c0[t] > 63) ok = false;
          }
          for (int dy = 0; dy < dh && ok; dy++) if (yofs[dy] < 0 || (dy > 0 && yofs[dy] <= yofs[dy - 1])) ok = false;
          for (int dx = 0; dx < 2 * dw && ok; dx++) if (ia[dx] < 0 || ia[dx] > 2048) ok = false;
          for (int dy = 0; dy < 2 * dh && ok; dy++) if (ib[dy] < 0 || ib[dy] > 2048) ok = false;
          if (ok) {
            std::vector<int8_t> W((size_t)nchunks * 6 * 64 * 16, 0);
            std::vector<int32_t> Cc((size_t)nchunks * 3 * 16, 0);
            for (int t = 0; t < nchunks; t++)
              for (int cb = 0; cb < 3; cb++)
                for (int m = 0; m < 16; m++) {
                  const int dx = 48 * t + 12 * (m >> 2) + 4 * cb + (m & 3);
                  if (dx >= dw) continue;
                  const int s0 = xofs[dx], s1 = std::min(xofs[dx] + 1, sw - 1);
                  const int wgt[2] = {ia[2 * dx], ia[2 * dx + 1]}, col[2] = {s0, s1};
                  Cc[((size_t)t * 3 + cb) * 16 + m] = 128 * (wgt[0] + wgt[1]);
                  for (int e = 0; e < 2; e++) {
                    const int k = col[e] - c0[t];                       // (0 .. 63: checked above)
                    const size_t lane = (size_t)m + 16 * (k >> 4), byte = (size_t)(k & 15);
                    int8_t* wh = &W[(((size_t)t * 6 + 2 * cb) * 64 + lane) * 16 + byte];
                    int8_t* wl = &W[(((size_t)t * 6 + 2 * cb + 1) * 64 + lane) * 16 + byte];
                    // (s1 == s0 at the right border: the two weights meet in one column and add up - a1 is 0 there)
                    const int tot = 32 * (int)*wh + (int)*wl + wgt[e];
                    *wh = (int8_t)(tot >> 5); *wl = (int8_t)(tot & 31);
                  }
                }
            const int smax = yofs[dh - 1], nblocks = (smax + 1 + 14) / 15;
            std::vector<uint32_t> rowtab(2 * ((size_t)15 * nblocks + 1), 0);
            for (size_t s2 = 0; s2 < rowtab.size() / 2; s2++) rowtab[2 * s2] = 0xFFFFFFFFu;        // dy = -1
            for (int dy = 0; dy < dh; dy++) { rowtab[2 * (size_t)yofs[dy]] = (uint32_t)dy; rowtab[2 * (size_t)yofs[dy] + 1] = (uint32_t)(uint16_t)ib[2 * dy] | ((uint32_t)(uint16_t)ib[2 * dy + 1] << 16); }
            M.oW = push(W.data(), W.size()); M.oC = push(Cc.data(), Cc.size() * 4); M.oC0 = push(c0.data(), c0.size() * 4); M.oRow = push(rowtab.data(), rowtab.size() * 4);
            M.nchunks = nchunks; M.nblocks = nblocks; M.ok = true;
          }
        }
      }
      // blur tiles
      for (int ty = 0; ty < (L.h + BLUR_TH - 1) / BLUR_TH; ty++)
        for (int tx = 0; tx < (L.w + BLUR_TW - 1) / BLUR_TW; tx++) {
          BlurTile bt; bt.level = (short)l; bt.tx = (short)tx; bt.ty = (short)ty; bt.pad = 0;
          c->btiles.push_back(bt);
        }
      for (int ty = 0, nty = (L.h + BM_TH - 1) / BM_TH; ty < nty; ty += BM_RC)       // k_blur7_mfma: ty = first 58-row chunk, pad = chunks of the workgroup
        for (int tx = 0; tx < (L.w + BM_TW - 1) / BM_TW; tx++) {
          BlurTile bt; bt.level = (short)l; bt.tx = (short)tx; bt.ty = (short)ty; bt.pad = (short)std::min(BM_RC, nty - ty);
          c->mtiles.push_back(bt);
          for (int q = 0; q < bt.pad; q++) { BlurTile b1 = bt; b1.ty = (short)(ty + q); b1.pad = 1; c->mtiles1.push_back(b1); }
        }
    }
    G.ncells_total = (int)c->cells.size();
    G.cell_cap = cell_cap; G.sel_cap = sel_cap; G.keys_per_frame = key_off; G.desc_blocks = desc_blocks;
    ORBHIP_REQUIRE(tile_w <= 64 && tile_h <= 64, ORBHIP_EINVAL, "FAST cell larger than 64 px (unsupported image geometry)");
    G.tile_w = tile_w; G.tile_h = tile_h; G.tile_pitch = round_up(tile_w, 4) + 4;
    G.node_cap = round_up(node_cap, 8); G.max_cells_level = round_up(max_cells, 8);
#ifdef ORBHIP_OCT_LEVEL_EXPERIMENT
    G.oct_level_mask = ORBHIP_EXP_ENV("ORBHIP_OCT_LEVELS") ? (int)strtol(ORBHIP_EXP_ENV("ORBHIP_OCT_LEVELS"), nullptr, 0) : 0xFFFF;
#endif
    // ---- k_pyr_cone: per 32 x 8 tile of the top level, the box it computes on every level (see the kernel)
    c->cone_wgs = 0;
    if (nl >= 3 && nl <= CONE_MAXL) {
      const int top = nl - 1, TW = 32, TH = 8;
      const int ntx = (G.lv[top].w + TW - 1) / TW, nty = (G.lv[top].h + TH - 1) / TH;
      std::vector<short> boxes((size_t)ntx * nty * nl * 4);
      int buf0 = 0, bufk = 0; size_t tabmax = 0; bool ok = G.lv[0].w < 32000 && G.lv[0].h < 32000;
      for (int j = 0; j < nty && ok; j++)
        for (int i = 0; i < ntx && ok; i++) {
          short* Bx = &boxes[((size_t)j * ntx + i) * nl * 4];
          int bx0 = i * TW, by0 = j * TH, bx1 = std::min((i + 1) * TW, G.lv[top].w), by1 = std::min((j + 1) * TH, G.lv[top].h);
          size_t tb = 0;
          for (int k = top; k >= 0; k--) {
            const int Wk = G.lv[k].w, Hk = G.lv[k].h;
            if (k < top) {
              // what the box of level k + 1 reads from level k ...
              const std::vector<int>& xo = h_xofs[k + 1]; const std::vector<int>& yo = h_yofs[k + 1];
              const int ux0 = Bx[4 * (k + 1)], ux1 = std::min<int>(Bx[4 * (k + 1) + 2], G.lv[k + 1].w), uy0 = Bx[4 * (k + 1) + 1], uy1 = Bx[4 * (k + 1) + 3];
              auto cy = [&](int v) { return std::min(std::max(v, 0), Hk - 1); };
              int nx0 = xo[ux0], nx1 = std::min(xo[ux1 - 1] + 1, Wk - 1) + 1, ny0 = cy(yo[uy0]), ny1 = cy(yo[uy1 - 1] + 1) + 1;
              for (int y = uy0; y < uy1; y++) { ny0 = std::min(ny0, cy(yo[y])); ny1 = std::max(ny1, cy(yo[y] + 1) + 1); }
              for (int x = ux0; x < ux1; x++) { nx0 = std::min(nx0, xo[x]); nx1 = std::max(nx1, std::min(xo[x] + 1, Wk - 1) + 1); }
              bx0 = nx0; bx1 = nx1; by0 = ny0; by1 = ny1;
              if (k >= 1) {                                       // ... and this workgroup's share of level k itself
                bx0 = std::min(bx0, (int)((long long)i * Wk / ntx)); bx1 = std::max(bx1, (int)((long long)(i + 1) * Wk / ntx));
                by0 = std::min(by0, (int)((long long)j * Hk / nty)); by1 = std::max(by1, (int)((long long)(j + 1) * Hk / nty));
              }
            }
            if (k >= 1) { bx0 &= ~3; bx1 = std::min(round_up(bx1, 4), round_up(Wk, 4)); }      // whole dwords, as k_resize stores them
            Bx[4 * k] = (short)bx0; Bx[4 * k + 1] = (short)by0; Bx[4 * k + 2] = (short)bx1; Bx[4 * k + 3] = (short)by1;
            const int bytes = (bx1 - bx0) * (by1 - by0);
            if (k >= 1 && (bx1 - bx0 > 256 || by1 - by0 > 256)) ok = false;      // (a thread loads one table entry per level)
            if (k == 0 && bytes > CONE_TPB * CONE_SRC_PT) ok = false;
            if (k == 0) buf0 = std::max(buf0, bytes); else { bufk = std::max(bufk, bytes); tb += 8 * (size_t)(bx1 - bx0) + 16 * (size_t)(by1 - by0); }
          }
          tabmax = std::max(tabmax, tb);
        }
      buf0 = round_up(buf0, 16); bufk = round_up(bufk, 16);
      const size_t lds = (size_t)buf0 + 2 * (size_t)bufk + tabmax + 64;
      if (ok && lds <= 96 * 1024) {
        size_t off = (tab.size() + 15) / 16 * 16;
        tab.resize(off + boxes.size() * 2);
        std::memcpy(tab.data() + off, boxes.data(), boxes.size() * 2);
        c->tab_cone = off; c->cone_wgs = ntx * nty; c->cone_buf0 = buf0; c->cone_bufk = bufk; c->cone_lds = lds;
      }
    }
    G.pyr_frame_bytes = (pyr_off + 255) / 256 * 256;
    G.blur_frame_bytes = (blur_off + 255) / 256 * 256;
    c->fast_narrow = tile_w - 6 <= 32;                      // every cell interior <= 32 px wide: k_fast_cells<true> (32-bit row masks)
    c->fast_lds = (size_t)round_up((int)((size_t)2 * round_up(G.tile_h * G.tile_pitch, 16) + 2 * 64 * (c->fast_narrow ? 4 : 8) + 16 + (size_t)2 * std::max(tile_w - 6, 1) * std::max(tile_h - 6, 1) + 16), 16);   // tile + score (u8) + row masks + queue counter + queue (u16)
    c->octree_wide = false;
    for (int l = 0; l < c->nlevels; l++) c->octree_wide = c->octree_wide || G.lv[l].kcap > 65535;
    c->octree_lds = octree_lds_bytes(G.node_cap, G.max_cells_level, false, false);
    c->octree_lds_wide = octree_lds_bytes(G.node_cap, G.max_cells_level, true, false);
    // node arrays beyond the LDS: both instantiations keep them in a global scratch row per (frame, level) instead (k_octree<.., true>)
    c->octree_gmem = (c->octree_wide ? c->octree_lds_wide : c->octree_lds) > 160 * 1024;
    c->octree_row = (size_t)round_up((int)octree_lds_bytes(G.node_cap, G.max_cells_level, true, true), 256);
    ORBHIP_REQUIRE(G.node_cap <= 32760, ORBHIP_EINVAL, "nfeatures too large: more than 32752 keypoints in one level (16-bit node indices)");
    if (int rc = c->d_cells.ensure(std::max<size_t>(c->cells.size(), 1) * sizeof(CellDesc))) return rc;
    if (int rc = c->d_btiles.ensure(c->btiles.size() * sizeof(BlurTile))) return rc;
    if (int rc = c->d_mtiles.ensure(c->mtiles.size() * sizeof(BlurTile))) return rc;
    if (int rc = c->d_mtiles1.ensure(c->mtiles1.size() * sizeof(BlurTile))) return rc;
    if (int rc = c->d_tab.ensure(std::max<size_t>(tab.size(), 16))) return rc;
    if (!c->cells.empty()) ORBHIP_CHECK_HIP(hipMemcpy(c->d_cells.p, c->cells.data(), c->cells.size() * sizeof(CellDesc), hipMemcpyHostToDevice));
    ORBHIP_CHECK_HIP(hipMemcpy(c->d_btiles.p, c->btiles.data(), c->btiles.size() * sizeof(BlurTile), hipMemcpyHostToDevice));
    ORBHIP_CHECK_HIP(hipMemcpy(c->d_mtiles.p, c->mtiles.data(), c->mtiles.size() * sizeof(BlurTile), hipMemcpyHostToDevice));
    ORBHIP_CHECK_HIP(hipMemcpy(c->d_mtiles1.p, c->mtiles1.data(), c->mtiles1.size() * sizeof(BlurTile), hipMemcpyHostToDevice));
    if (!tab.empty()) ORBHIP_CHECK_HIP(hipMemcpy(c->d_tab.p, tab.data(), tab.size(), hipMemcpyHostToDevice));
    if (!c->octree_gmem && c->octree_lds > 64 * 1024)
      if (int rc = raise_dynamic_lds((const void*)k_octree<false, false>, c->device, c->octree_lds)) return rc;
    if (!c->octree_gmem && c->octree_wide && c->octree_lds_wide > 64 * 1024)
      if (int rc = raise_dynamic_lds((const void*)k_octree<true, false>, c->device, c->octree_lds_wide)) return rc;
    c->w = w; c->h = h; c->stride = stride; c->nframes = 0;
  }
  if (!c->const_uploaded) {
    ORBHIP_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_pattern), ORB_BIT_PATTERN_31, 1024));
    ORBHIP_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_umax), c->umax.data(), 16 * sizeof(int)));
    {
      // k_blur7_mfma's Toeplitz operands (see the kernel): lane (n, g), byte j <-> k = 16 g + j
      std::vector<int8_t> T((size_t)2 * 7 * 64 * 16, 0);
      for (int var = 0; var < 2; var++) {
        const int taps[7] = {18, 34, var ? 48 : 49, var ? 56 : 55, var ? 48 : 49, 34, 18};
        for (int q = 0; q < 7; q++)
          for (int lane = 0; lane < 64; lane++)
            for (int j = 0; j < 16; j++) {
              const int n = lane & 15, kk = 16 * (lane >> 4) + j;
              const int ti = q < 3 ? kk - (12 * (n >> 2) + 4 * q + (n & 3)) - 1      // row pass, block q: output column c + 12 (n >> 2) + 4 q + (n & 3), source column c - 4 + kk
                                   : kk - (16 * (q - 3) + n);                         // column pass, block s = q - 3: output row R0 + 3 + 16 s + n, source row R0 + kk
              T[(((size_t)var * 7 + q) * 64 + lane) * 16 + j] = (int8_t)((ti >= 0 && ti <= 6) ? taps[ti] : 0);
            }
      }
      ORBHIP_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_blur_toep), T.data(), T.size()));
    }
    c->const_uploaded = true;
  }
  const size_t B = (size_t)nframes;
  if (int rc = c->d_pyr.ensure(std::max<size_t>(B * G.pyr_frame_bytes, 256))) return rc;
  if (int rc = c->d_blur.ensure(B * G.blur_frame_bytes)) return rc;
  if (int rc = c->d_cellcnt.ensure(std::max<size_t>(B * G.ncells_total * 4, 16))) return rc;
  if (int rc = c->d_cellkps.ensure(std::max<size_t>(B * G.ncells_total * (size_t)G.cell_cap * 4, 16))) return rc;
  if (int rc = c->d_keys.ensure(B * G.keys_per_frame * 4)) return rc;
  if (int rc = c->d_knode.ensure(B * G.keys_per_frame * 2)) return rc;
  if (int rc = c->d_sel.ensure(B * nl * G.sel_cap * 4)) return rc;
  if (c->octree_gmem) { if (int rc = c->d_octnodes.ensure(B * nl * c->octree_row)) return rc; }
  if (int rc = c->d_selcnt.ensure(B * nl * 4)) return rc;
  if (int rc = c->d_nkeys.ensure(B * nl * 4)) return rc;
  if (int rc = c->d_status.ensure(B * 4)) return rc;
  c->nframes = nframes;
  return 0;
}

static int run_batch(orbx_ctx* c, const uint8_t* d_imgs, int w, int h, int stride, size_t frame_stride,
                     int nframes, orbx_keypoint* d_kps, uint8_t* d_desc, int cap, int32_t* d_counts,
                     hipStream_t st) {
  void (*blur_k)(GeomDev, const BlurTile*, const uint8_t*, long long, const uint8_t*, uint8_t*) =
      c->blur_mfma ? (c->blur_variant ? k_blur7_mfma<1> : k_blur7_mfma<0>) : (c->blur_variant ? k_blur7<1> : k_blur7<0>);

  ORBHIP_CHECK_HIP(hipSetDevice(c->device));
  if (int rc = prepare(c, w, h, stride, nframes)) return rc;
  const GeomDev& G = c->G;
  const int nl = c->nlevels;
  uint8_t* pyr = c->d_pyr.as<uint8_t>();
  // (a lone frame: one 58-row chunk per wave - four times the workgroups, a quarter of the chain per wave)
  const bool blur_short = c->blur_mfma && nframes < 4;
  const unsigned n_btiles = (unsigned)(c->blur_mfma ? (blur_short ? c->mtiles1.size() : c->mtiles.size()) : c->btiles.size());
  const BlurTile* d_btl = c->blur_mfma ? (blur_short ? c->d_mtiles1.as<BlurTile>() : c->d_mtiles.as<BlurTile>()) : c->d_btiles.as<BlurTile>();
  auto mark = [&]() { if (c->profiling) { hipEvent_t e; if (hipEventCreate(&e) == hipSuccess) { (void)hipEventRecord(e, st); c->prof_events.push_back(e); } } };
  static const bool cone_on = []() { const char* e = ORBHIP_EXP_ENV("ORBHIP_EXTRACT_CONE"); return !(e && e[0] == '0'); }();
  const bool cone = cone_on && nframes == 1 && c->cone_wgs > 0;
  if (!cone) ORBHIP_CHECK_HIP(hipMemsetAsync(c->d_status.p, 0, (size_t)nframes * 4, st));
  mark();
  // pyramid chain: a single frame takes the one-launch cone kernel (latency), batches one launch per level (throughput;
  // ORBHIP_EXTRACT_CONE=0: always per level)
  if (cone) {
    ConeArgs ca; std::memset(&ca, 0, sizeof(ca));
    const uint8_t* T = c->d_tab.as<uint8_t>();
    ca.nl = nl; ca.spitch0 = G.lv[0].pitch; ca.buf0 = c->cone_buf0; ca.bufk = c->cone_bufk;
    for (int l = 1; l < nl; l++) {
      ConeLevel& L = ca.lv[l];
      L.sw = G.lv[l - 1].w; L.sh = G.lv[l - 1].h; L.dw = G.lv[l].w; L.dh = G.lv[l].h; L.dpitch = G.lv[l].pitch; L.doff = G.lv[l].pyr_off;
      L.xtab = (const uint2*)(T + c->tab_xofs[l]); L.yofs = (const int*)(T + c->tab_yofs[l]); L.ibeta = (const short*)(T + c->tab_ibeta[l]);
    }
    if (c->cone_lds > 64 * 1024)
      if (int rc = raise_dynamic_lds((const void*)k_pyr_cone, c->device, c->cone_lds)) return rc;
    hipLaunchKernelGGL(k_pyr_cone, dim3(c->cone_wgs), dim3(CONE_TPB), c->cone_lds, st, ca, (const short*)(T + c->tab_cone), d_imgs, pyr, c->d_status.as<int>());
  }
  for (int l = 1; l < nl && !cone; l++) {
    const LevelDev& S = G.lv[l - 1];
    const LevelDev& D = G.lv[l];
    const uint8_t* src = (l == 1) ? d_imgs : pyr + S.pyr_off;
    long long sframe = (l == 1) ? (long long)frame_stride : G.pyr_frame_bytes;
    const uint8_t* T = c->d_tab.as<uint8_t>();
    if (c->resize_mfma && (size_t)l < c->rm.size() && c->rm[l].ok && nframes <= 65535) {
      const orbx_ctx::RmHost& M = c->rm[l];
      RmLevel R; R.tab = T; R.oW = M.oW; R.oC = M.oC; R.oC0 = M.oC0; R.oRow = M.oRow; R.nchunks = M.nchunks; R.nblocks = M.nblocks; R.sw = S.w; R.sh = S.h; R.dw = D.w; R.dh = D.h;
      const int ntx = (M.nchunks + 3) / 4, ntb = (M.nblocks + RM_RB - 1) / RM_RB;
      hipLaunchKernelGGL(k_resize_mfma, dim3(ntx * ntb, nframes), dim3(256), 0, st, R, src, S.pitch, sframe, pyr + D.pyr_off, D.pitch, G.pyr_frame_bytes, ntx);
      continue;
    }
    dim3 grid((D.w + 255) / 256, (D.h + 4 * RS_ROWS - 1) / (4 * RS_ROWS), nframes), block(64, 4);
    hipLaunchKernelGGL(k_resize, grid, block, 0, st, src, S.pitch, sframe, S.w, S.h, pyr + D.pyr_off, D.pitch,
                       G.pyr_frame_bytes, D.w, D.h, (const uint2*)(T + c->tab_xofs[l]), (const int*)(T + c->tab_yofs[l]),
                       (const short*)(T + c->tab_ibeta[l]));
  }
  // (default -1: batches of >= 8 frames run the blur beside FAST + octree - the octree is a handful of long workgroups that leave
  // most of the chip idle: +3.2 % on the one-stream bench, 112.1k -> 115.5k frames/s; starting the blur of level 0 even earlier,
  // beside the resize chain, was measured too and adds nothing.  A lone frame stays on one stream: a fork / join costs more
  // than it hides)
  // (a lone frame inside a longer device chain - the Tracking step - does fork: the host is ahead of the device there, so the
  // fork / join costs nothing on the critical path and the blur's 13.6 us run beside FAST + octree: 0.310 -> 0.297 ms per step)
  int side_mode = c->ev_fork ? (c->overlap_blur >= 0 ? c->overlap_blur : (nframes >= 8 ? 1 : (nframes == 1 ? c->lone_side_mode : 0))) : 0;
  // The blur of a lone frame inside the per-frame Tracking chain is on that frame's critical path, and the Tracking thread may run
  // beside another thread's bundle adjustment (orbhip_set_thread_priority): it gets a stream of the greatest priority.  Batches do
  // NOT (round 5): a prioritised blur starves the latency-bound octree of its own batch (k_octree 0.40 -> 0.55 ms) and
  // the copy / compute overlap of a host-fed pipeline (112 k -> 73 k frames/s).  ORBHIP_SIDE_PRIORITY=1 restores round 4's behaviour.
  // Either stream is created on first use: streams that merely exist cost dispatch slots.
  static const bool side_prio_all = []() { const char* v = std::getenv("ORBHIP_SIDE_PRIORITY"); return v && v[0] == '1'; }();
  hipStream_t side_st = nullptr;
  if (side_mode) {
    const bool want_hi = (nframes == 1 && c->lone_side_mode) || side_prio_all;
    if (want_hi && !c->side_hi) {
      int lo = 0, hi = 0;
      if (!(hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi != lo && hipStreamCreateWithPriority(&c->side_hi, hipStreamNonBlocking, hi) == hipSuccess)) c->side_hi = nullptr;
    }
    if (want_hi && c->side_hi) side_st = c->side_hi;
    else {
      if (!c->side && hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess) c->side = nullptr;
      side_st = c->side;
    }
    if (!side_st) side_mode = 0;
  }
  c->lone_side_mode = 0;
  auto launch_blur_side = [&]() -> int {
    ORBHIP_CHECK_HIP(hipEventRecord(c->ev_fork, st));
    ORBHIP_CHECK_HIP(hipStreamWaitEvent(side_st, c->ev_fork, 0));
    hipEvent_t sb = nullptr, se = nullptr;
    if (c->profiling) { (void)hipEventCreate(&sb); (void)hipEventCreate(&se); (void)hipEventRecord(sb, side_st); }
    hipLaunchKernelGGL(blur_k, dim3(n_btiles, nframes), dim3(256), 0, side_st, G,
                       d_btl, d_imgs, (long long)frame_stride, pyr, c->d_blur.as<uint8_t>());
    if (c->profiling) { (void)hipEventRecord(se, side_st); c->side_events.push_back(sb); c->side_events.push_back(se); }
    ORBHIP_CHECK_HIP(hipEventRecord(c->ev_join, side_st));
    return 0;
  };
  if (side_mode == 1) { if (int rc = launch_blur_side()) return rc; }
  mark();
  if (G.ncells_total > 0) {
    if (c->fast_narrow)
      hipLaunchKernelGGL(k_fast_cells<true>, dim3((G.ncells_total + FAST_WPB * FAST_CPW - 1) / (FAST_WPB * FAST_CPW), nframes), dim3(64 * FAST_WPB), c->fast_lds * FAST_WPB, st, G,
                         c->d_cells.as<CellDesc>(), d_imgs, (long long)frame_stride, pyr, c->d_cellcnt.as<int>(),
                         c->d_cellkps.as<uint32_t>(), c->iniTh, c->minTh, (int)c->fast_lds, (c->fast_xcd && nframes % 8 == 0) ? 1 : 0);
    else
      hipLaunchKernelGGL(k_fast_cells<false>, dim3((G.ncells_total + FAST_WPB * FAST_CPW - 1) / (FAST_WPB * FAST_CPW), nframes), dim3(64 * FAST_WPB), c->fast_lds * FAST_WPB, st, G,
                         c->d_cells.as<CellDesc>(), d_imgs, (long long)frame_stride, pyr, c->d_cellcnt.as<int>(),
                         c->d_cellkps.as<uint32_t>(), c->iniTh, c->minTh, (int)c->fast_lds, (c->fast_xcd && nframes % 8 == 0) ? 1 : 0);
  }
  mark();
  // mode 2: the octree is one latency-bound workgroup per (frame, level) - 6 % VALU-busy, 0.6 waves per SIMD - so the
  // VALU-bound blur runs BESIDE it: the fork is taken after FAST, the octree is submitted first and keeps its slots
  if (side_mode == 2) ORBHIP_CHECK_HIP(hipEventRecord(c->ev_fork, st));
  if (!c->octree_gmem && c->octree_wide && nframes == 1) {
    const size_t lds2 = std::max(c->octree_lds, c->octree_lds_wide);
    if (lds2 > 64 * 1024)
      if (int rc = raise_dynamic_lds((const void*)k_octree_pair, c->device, lds2)) return rc;
    hipLaunchKernelGGL(k_octree_pair, dim3(2 * nl, 1), dim3(OCT_TPB), lds2, st, G, c->d_cellcnt.as<int>(), c->d_cellkps.as<uint32_t>(), c->d_keys.as<uint32_t>(),
                       c->d_knode.as<unsigned short>(), c->d_sel.as<uint32_t>(), c->d_selcnt.as<int>(), c->d_nkeys.as<int>(), c->d_status.as<int>());
  } else if (!c->octree_gmem) {
    hipLaunchKernelGGL((k_octree<false, false>), dim3(nl, nframes), dim3(OCT_TPB), c->octree_lds, st, G, c->d_cellcnt.as<int>(),
                       c->d_cellkps.as<uint32_t>(), c->d_keys.as<uint32_t>(), c->d_knode.as<unsigned short>(),
                       c->d_sel.as<uint32_t>(), c->d_selcnt.as<int>(), c->d_nkeys.as<int>(), c->d_status.as<int>(), (uint8_t*)nullptr, (size_t)0);
    if (c->octree_wide)        // levels with more than 65535 candidates (32-bit node counters); every other workgroup leaves at once
      hipLaunchKernelGGL((k_octree<true, false>), dim3(nl, nframes), dim3(OCT_TPB), c->octree_lds_wide, st, G, c->d_cellcnt.as<int>(),
                         c->d_cellkps.as<uint32_t>(), c->d_keys.as<uint32_t>(), c->d_knode.as<unsigned short>(),
                         c->d_sel.as<uint32_t>(), c->d_selcnt.as<int>(), c->d_nkeys.as<int>(), c->d_status.as<int>(), (uint8_t*)nullptr, (size_t)0);
  } else {                     // per-level quota beyond the LDS: node arrays in a global scratch row per (frame, level)
    hipLaunchKernelGGL((k_octree<false, true>), dim3(nl, nframes), dim3(OCT_TPB), 0, st, G, c->d_cellcnt.as<int>(),
                       c->d_cellkps.as<uint32_t>(), c->d_keys.as<uint32_t>(), c->d_knode.as<unsigned short>(),
                       c->d_sel.as<uint32_t>(), c->d_selcnt.as<int>(), c->d_nkeys.as<int>(), c->d_status.as<int>(), c->d_octnodes.as<uint8_t>(), c->octree_row);
    if (c->octree_wide)
      hipLaunchKernelGGL((k_octree<true, true>), dim3(nl, nframes), dim3(OCT_TPB), 0, st, G, c->d_cellcnt.as<int>(),
                         c->d_cellkps.as<uint32_t>(), c->d_keys.as<uint32_t>(), c->d_knode.as<unsigned short>(),
                         c->d_sel.as<uint32_t>(), c->d_selcnt.as<int>(), c->d_nkeys.as<int>(), c->d_status.as<int>(), c->d_octnodes.as<uint8_t>(), c->octree_row);
  }
  if (side_mode == 2) {
    ORBHIP_CHECK_HIP(hipStreamWaitEvent(side_st, c->ev_fork, 0));
    hipEvent_t sb = nullptr, se = nullptr;
    if (c->profiling) { (void)hipEventCreate(&sb); (void)hipEventCreate(&se); (void)hipEventRecord(sb, side_st); }
    hipLaunchKernelGGL(blur_k, dim3(n_btiles, nframes), dim3(256), 0, side_st, G,
                       d_btl, d_imgs, (long long)frame_stride, pyr, c->d_blur.as<uint8_t>());
    if (c->profiling) { (void)hipEventRecord(se, side_st); c->side_events.push_back(sb); c->side_events.push_back(se); }
    ORBHIP_CHECK_HIP(hipEventRecord(c->ev_join, side_st));
  }
  mark();
  if (side_mode == 0) hipLaunchKernelGGL(blur_k, dim3(n_btiles, nframes), dim3(256), 0, st, G,
                                         d_btl, d_imgs, (long long)frame_stride, pyr, c->d_blur.as<uint8_t>());
  else ORBHIP_CHECK_HIP(hipStreamWaitEvent(st, c->ev_join, 0));       // join: describe needs the blurred levels
  mark();
  hipLaunchKernelGGL(k_describe, dim3(G.desc_blocks, nframes), dim3(64 * DESC_WPB), 0, st, G, c->d_sel.as<uint32_t>(),
                     c->d_selcnt.as<int>(), c->d_status.as<int>(), d_imgs, (long long)frame_stride, pyr,
                     c->d_blur.as<uint8_t>(), d_kps, d_desc, cap, d_counts, c->atan_p[0], c->atan_p[1],
                     c->atan_p[2], c->atan_p[3], c->factorPI, (c->desc_xcd && nframes % 8 == 0) ? 1 : 0);
  mark();
  ORBHIP_CHECK_HIP(hipGetLastError());
  c->last_img0 = d_imgs; c->last_img_frame_bytes = (long long)frame_stride; c->last_nframes = nframes;
  return 0;
}

extern "C" {

int orbx_create(int nfeatures, float scale_factor, int nlevels, int ini_th_fast, int min_th_fast, int device,
                orbx_ctx** out) {
  ORBHIP_REQUIRE(out != nullptr, ORBHIP_EINVAL, "out is NULL");
  ORBHIP_REQUIRE(nfeatures > 0 && nlevels >= 1 && nlevels <= MAX_LEVELS && scale_factor > 1.0f, ORBHIP_EINVAL,
                 "bad extractor parameters");
  ORBHIP_REQUIRE(min_th_fast >= 1 && ini_th_fast >= min_th_fast && ini_th_fast <= 254, ORBHIP_EINVAL,
                 "FAST thresholds must satisfy 1 <= min <= ini <= 254");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device available (the HIP path has no CPU fallback)"); return ORBHIP_ENODEV; }
  ORBHIP_REQUIRE(device >= 0 && device < ndev, ORBHIP_EINVAL, "device ordinal out of range");
  orbx_ctx* c = new (std::nothrow) orbx_ctx();
  ORBHIP_REQUIRE(c != nullptr, ORBHIP_ENOMEM, "out of host memory");
  c->nfeatures = nfeatures; c->scaleFactor = scale_factor; c->nlevels = nlevels;
  c->iniTh = ini_th_fast; c->minTh = min_th_fast; c->device = device;
  build_tables(c);
  (void)hipSetDevice(device);
  // (fork / join events of the blur's side stream; the streams themselves are created by run_batch on first use)
  if (hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess) c->ev_fork = nullptr;
  if (c->ev_fork && hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) { (void)hipEventDestroy(c->ev_fork); c->ev_fork = nullptr; c->ev_join = nullptr; }
  if (const char* e = std::getenv("ORBHIP_BLUR_MFMA")) c->blur_mfma = atoi(e) != 0;
  if (const char* e = std::getenv("ORBHIP_RESIZE_MFMA")) c->resize_mfma = atoi(e) != 0;
  if (const char* e = ORBHIP_EXP_ENV("ORBHIP_OVERLAP_BLUR")) c->overlap_blur = atoi(e);
  if (const char* e = ORBHIP_EXP_ENV("ORBHIP_DESC_XCD")) c->desc_xcd = atoi(e);
  if (const char* e = ORBHIP_EXP_ENV("ORBHIP_FAST_XCD")) c->fast_xcd = atoi(e);
  *out = c;
  return 0;
}

int orbx_destroy(orbx_ctx* c) {
  if (!c) return 0;
  DevBuf* bufs[] = {&c->d_cells, &c->d_btiles, &c->d_mtiles, &c->d_mtiles1, &c->d_tab, &c->d_pyr, &c->d_blur, &c->d_cellcnt, &c->d_cellkps,
                    &c->d_keys, &c->d_knode, &c->d_sel, &c->d_selcnt, &c->d_nkeys, &c->d_status, &c->d_img,
                    &c->d_out, &c->d_octnodes};
  for (DevBuf* b : bufs) b->release();
  if (c->h_pin) (void)hipHostFree(c->h_pin);
  if (c->side) { (void)hipStreamSynchronize(c->side); (void)hipStreamDestroy(c->side); }
  if (c->side_hi) { (void)hipStreamSynchronize(c->side_hi); (void)hipStreamDestroy(c->side_hi); }
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  delete c;
  return 0;
}

#ifdef ORBHIP_OCT_PROF
int orbx_debug_oct_prof(unsigned long long* out /*[MAX_LEVELS][16]*/, int reset) {
  ORBHIP_CHECK_HIP(hipDeviceSynchronize());
  ORBHIP_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_oct_prof), sizeof(unsigned long long) * MAX_LEVELS * 16));
  if (reset) { std::vector<unsigned long long> z((size_t)MAX_LEVELS * 16, 0ull); ORBHIP_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_oct_prof), z.data(), z.size() * 8)); }
  return 0;
}
#endif

#ifdef ORBHIP_FAST_PROF
int orbx_debug_fast_prof(unsigned long long* out8) {          // sums over the waves of the LAST launch: ticks per phase, [7] = waves
  ORBHIP_CHECK_HIP(hipDeviceSynchronize());
  std::vector<unsigned int> h((size_t)FAST_PROF_WAVES * 8);
  ORBHIP_CHECK_HIP(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_fast_prof), h.size() * 4));
  for (int k = 0; k < 8; k++) out8[k] = 0;
  for (size_t w = 0; w < FAST_PROF_WAVES; w++) if (h[8 * w + 7]) { for (int k = 0; k < 5; k++) out8[k] += h[8 * w + k]; out8[7]++; }
  return 0;
}
#endif

int orbx_set_profiling(orbx_ctx* c, int enable) {
  ORBHIP_REQUIRE(c != nullptr, ORBHIP_EINVAL, "ctx is NULL");
  for (hipEvent_t e : c->prof_events) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->side_events) (void)hipEventDestroy(e);
  c->prof_events.clear(); c->side_events.clear();
  c->profiling = enable != 0;
  return 0;
}

int orbx_get_stage_ms(orbx_ctx* c, float* ms, int* ncalls) {
  ORBHIP_REQUIRE(c && ms && ncalls, ORBHIP_EINVAL, "NULL argument");
  ORBHIP_CHECK_HIP(hipSetDevice(c->device));
  for (int s = 0; s < 5; s++) ms[s] = 0.f;
  const size_t n = c->prof_events.size() / 6;
  if (n) ORBHIP_CHECK_HIP(hipEventSynchronize(c->prof_events.back()));
  for (size_t k = 0; k < n; k++)
    for (int s = 0; s < 5; s++) {
      float t = 0.f;
      ORBHIP_CHECK_HIP(hipEventElapsedTime(&t, c->prof_events[6 * k + s], c->prof_events[6 * k + s + 1]));
      ms[s] += t;
    }
  if (c->side_events.size() == 2 * n && n) {          // blur ran on the side stream: its own duration, not the join wait
    ms[3] = 0.f;
    ORBHIP_CHECK_HIP(hipEventSynchronize(c->side_events.back()));
    for (size_t k = 0; k < n; k++) { float t = 0.f; ORBHIP_CHECK_HIP(hipEventElapsedTime(&t, c->side_events[2 * k], c->side_events[2 * k + 1])); ms[3] += t; }
  }
  *ncalls = (int)n;
  for (hipEvent_t e : c->prof_events) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->side_events) (void)hipEventDestroy(e);
  c->prof_events.clear(); c->side_events.clear();
  return 0;
}

int orbx_get_levels(const orbx_ctx* c) { return c ? c->nlevels : ORBHIP_EINVAL; }
int orbx_set_opencv_variant(orbx_ctx* c, int blur_variant) {
  ORBHIP_REQUIRE(c && (blur_variant == ORBX_CV_BLUR_8BIT || blur_variant == ORBX_CV_BLUR_FIXED16), ORBHIP_EINVAL, "unknown OpenCV variant");
  c->blur_variant = blur_variant;
  return 0;
}

int orbx_get_tables(const orbx_ctx* c, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                    int32_t* fpl) {
  ORBHIP_REQUIRE(c != nullptr, ORBHIP_EINVAL, "ctx is NULL");
  for (int i = 0; i < c->nlevels; i++) {
    if (scale) scale[i] = c->scale[i];
    if (inv_scale) inv_scale[i] = c->inv_scale[i];
    if (sigma2) sigma2[i] = c->sigma2[i];
    if (inv_sigma2) inv_sigma2[i] = c->inv_sigma2[i];
    if (fpl) fpl[i] = c->quota[i];
  }
  return 0;
}

int orbx_max_keypoints(const orbx_ctx* c) {
  if (!c) return ORBHIP_EINVAL;
  // DistributeOctTree returns at most N + 3 keypoints per level once it is past its first sweep, but that first sweep splits
  // all nIni initial nodes unconditionally (src/ORBextractor.cc:589-660): a wide, short level with few features can come
  // back with up to 4 * nIni > N keypoints.  nIni depends on the image aspect ratio (<= MAX_INI), so the bound is taken over it.
  int s = 0;
  for (int q : c->quota) s += std::max(q + 3, 4 * MAX_INI);
  return s;
}

int orbx_extract_batch_device(orbx_ctx* c, const uint8_t* d_imgs, int w, int h, int stride, size_t frame_stride,
                              int nframes, orbx_keypoint* d_kps, uint8_t* d_desc, int cap, int32_t* d_counts,
                              void* stream) {
  ORBHIP_REQUIRE(c && d_imgs && d_kps && d_desc && d_counts, ORBHIP_EINVAL, "NULL argument");
  ORBHIP_REQUIRE(w > 0 && h > 0 && stride >= w && nframes > 0 && cap > 0, ORBHIP_EINVAL, "bad dimensions");
  ORBHIP_REQUIRE(nframes <= 65535, ORBHIP_EINVAL, "at most 65535 frames per batch");
  return run_batch(c, d_imgs, w, h, stride, frame_stride, nframes, d_kps, d_desc, cap, d_counts, (hipStream_t)stream);
}

}  // extern "C"
namespace orbhip {
// orbx_extract_batch_device for ONE frame as a link of a longer device-resident chain (orb_track.hip): the blur goes to the
// extractor's side stream.  Not part of the C ABI.
int orbx_ctx_device(const orbx_ctx* c) { return c ? c->device : -1; }
unsigned long long orbx_ctx_generation(const orbx_ctx* c) { return c ? c->generation : 0ull; }
int orbx_extract_chained(orbx_ctx* c, const uint8_t* d_img, int w, int h, int stride, orbx_keypoint* d_kps, uint8_t* d_desc, int cap,
                         int32_t* d_count, void* stream) {
  if (c) c->lone_side_mode = 1;
  return orbx_extract_batch_device(c, d_img, w, h, stride, (size_t)stride * h, 1, d_kps, d_desc, cap, d_count, stream);
}
}  // namespace orbhip
extern "C" {

// Single-frame host path (what Frame::ExtractORB -> operator() is in the reference): latency matters more than bandwidth
// here.  The image goes up with ONE 1-D copy through a pinned staging buffer (a 2-D pageable copy of a 1241-byte-stride
// image cost 2.5 ms on its own) and is used with the caller's stride; counts, keypoints and descriptors live in ONE
// device block and come back with ONE copy into pinned memory.
int orbx_extract(orbx_ctx* c, const uint8_t* img, int w, int h, int stride, orbx_keypoint* kps, uint8_t* desc32,
                 int cap, int* n) {
  ORBHIP_REQUIRE(c != nullptr, ORBHIP_EINVAL, "ctx is NULL");
  if (!img || w <= 0 || h <= 0) { if (n) *n = 0; return 0; }    // empty image: silent return (:1046), nothing written
  ORBHIP_REQUIRE(kps && desc32 && n && cap > 0 && stride >= w, ORBHIP_EINVAL, "bad argument");
  ORBHIP_CHECK_HIP(hipSetDevice(c->device));
  const int icap = orbx_max_keypoints(c);
  const size_t img_bytes = (size_t)stride * (h - 1) + w;        // the caller owns exactly this span
  const size_t off_kps = 64, off_desc = off_kps + (size_t)icap * sizeof(orbx_keypoint), out_bytes = off_desc + (size_t)icap * 32;
  if (int rc = c->d_img.ensure((size_t)stride * h + 64)) return rc;
  if (int rc = c->d_out.ensure(out_bytes)) return rc;
  if (c->h_bytes < std::max(img_bytes, out_bytes)) {
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    c->h_pin = nullptr; c->h_bytes = 0;
    const size_t want = std::max(img_bytes, out_bytes) * 5 / 4 + 4096;
    ORBHIP_CHECK_HIP(hipHostMalloc(&c->h_pin, want, hipHostMallocDefault));
    c->h_bytes = want;
  }
  std::memcpy(c->h_pin, img, img_bytes);
  ORBHIP_CHECK_HIP(orbhip::ws_copy(c->d_img.p, c->h_pin, img_bytes, hipMemcpyHostToDevice, 0));        // (copy kernel on pinned staging: common.h)
  uint8_t* dout = c->d_out.as<uint8_t>();
  if (int rc = run_batch(c, c->d_img.as<uint8_t>(), w, h, stride, (size_t)stride * h, 1, (orbx_keypoint*)(dout + off_kps), dout + off_desc, icap,
                         (int32_t*)dout, 0))
    return rc;
  ORBHIP_CHECK_HIP(orbhip::ws_copy(c->h_pin, dout, out_bytes, hipMemcpyDeviceToHost, 0));
  ORBHIP_CHECK_HIP(hipStreamSynchronize(0));
  const int32_t cnt = *(const int32_t*)c->h_pin;
  if (cnt == -1) { set_error("candidate capacity exceeded (ORBHIP_KEYCAP test hook, or more than %d FAST corners in one pyramid level)", KEYCAP_MAX); return ORBHIP_EOVERFLOW; }
  ORBHIP_REQUIRE(cnt >= 0, ORBHIP_EOVERFLOW, "internal keypoint capacity exceeded");
  ORBHIP_REQUIRE(cnt <= cap, ORBHIP_ECAP, "output capacity too small");
  if (cnt > 0) {
    std::memcpy(kps, (const uint8_t*)c->h_pin + off_kps, (size_t)cnt * sizeof(orbx_keypoint));
    std::memcpy(desc32, (const uint8_t*)c->h_pin + off_desc, (size_t)cnt * 32);
  }
  *n = cnt;
  return 0;
}

int orbx_get_level_image(orbx_ctx* c, int frame, int level, int blurred, uint8_t* out, int* w, int* h) {
  ORBHIP_REQUIRE(c && c->last_nframes > 0, ORBHIP_EINVAL, "no extract call yet");
  ORBHIP_REQUIRE(frame >= 0 && frame < c->last_nframes && level >= 0 && level < c->nlevels, ORBHIP_EINVAL, "bad index");
  const LevelDev& L = c->G.lv[level];
  if (w) *w = L.w;
  if (h) *h = L.h;
  if (!out) return 0;
  ORBHIP_CHECK_HIP(hipSetDevice(c->device));
  ORBHIP_CHECK_HIP(hipDeviceSynchronize());
  const uint8_t* src; int pitch;
  if (blurred) { src = c->d_blur.as<uint8_t>() + (long long)frame * c->G.blur_frame_bytes + L.blur_off; pitch = L.bpitch; }
  else if (level == 0) { src = c->last_img0 + (long long)frame * c->last_img_frame_bytes; pitch = L.pitch; }
  else { src = c->d_pyr.as<uint8_t>() + (long long)frame * c->G.pyr_frame_bytes + L.pyr_off; pitch = L.pitch; }
  ORBHIP_CHECK_HIP(hipMemcpy2D(out, L.w, src, pitch, L.w, L.h, hipMemcpyDeviceToHost));
  return 0;
}

static void unpack_keys(const std::vector<uint32_t>& k, int32_t* out, int cap) {
  for (size_t i = 0; i < k.size() && (int)i < cap; i++) {
    out[3 * i] = (int32_t)(k[i] & 0xFFF); out[3 * i + 1] = (int32_t)((k[i] >> 12) & 0xFFF); out[3 * i + 2] = (int32_t)(k[i] >> 24);
  }
}

int orbx_get_level_candidates(orbx_ctx* c, int frame, int level, int32_t* out, int cap, int* n) {
  ORBHIP_REQUIRE(c && c->last_nframes > 0 && n, ORBHIP_EINVAL, "no extract call yet");
  ORBHIP_REQUIRE(frame >= 0 && frame < c->last_nframes && level >= 0 && level < c->nlevels, ORBHIP_EINVAL, "bad index");
  ORBHIP_CHECK_HIP(hipSetDevice(c->device));
  ORBHIP_CHECK_HIP(hipDeviceSynchronize());
  int nk = 0;
  ORBHIP_CHECK_HIP(hipMemcpy(&nk, c->d_nkeys.as<int>() + frame * c->nlevels + level, 4, hipMemcpyDeviceToHost));
  *n = nk;
  if (!out || nk <= 0) return 0;
  ORBHIP_REQUIRE(nk <= c->G.lv[level].kcap, ORBHIP_EOVERFLOW, "candidate capacity exceeded");
  std::vector<uint32_t> k(nk);
  ORBHIP_CHECK_HIP(hipMemcpy(k.data(), c->d_keys.as<uint32_t>() + (long long)frame * c->G.keys_per_frame + c->G.lv[level].key_off,
                             (size_t)nk * 4, hipMemcpyDeviceToHost));
  unpack_keys(k, out, cap);
  return 0;
}

int orbx_get_level_selected(orbx_ctx* c, int frame, int level, int32_t* out, int cap, int* n) {
  ORBHIP_REQUIRE(c && c->last_nframes > 0 && n, ORBHIP_EINVAL, "no extract call yet");
  ORBHIP_REQUIRE(frame >= 0 && frame < c->last_nframes && level >= 0 && level < c->nlevels, ORBHIP_EINVAL, "bad index");
  ORBHIP_CHECK_HIP(hipSetDevice(c->device));
  ORBHIP_CHECK_HIP(hipDeviceSynchronize());
  int nk = 0;
  ORBHIP_CHECK_HIP(hipMemcpy(&nk, c->d_selcnt.as<int>() + frame * c->nlevels + level, 4, hipMemcpyDeviceToHost));
  *n = nk;
  if (!out || nk <= 0) return 0;
  nk = std::min(nk, c->G.sel_cap);
  std::vector<uint32_t> k(nk);
  ORBHIP_CHECK_HIP(hipMemcpy(k.data(), c->d_sel.as<uint32_t>() + ((long long)frame * c->nlevels + level) * c->G.sel_cap,
                             (size_t)nk * 4, hipMemcpyDeviceToHost));
  unpack_keys(k, out, cap);
  return 0;
}

}  // extern "C"

#ifdef ORBHIP_CONE_PROF
extern "C" int orbx_debug_cone_ticks(unsigned long long* out) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cone_ticks), sizeof(unsigned long long) * 24) == hipSuccess ? 0 : -1;
}
#endif
