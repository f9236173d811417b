// Backward pass of a network's FIRST conv block -- 3x3 convolution of a <= 4-channel NHWC4 input, BatchNorm, leaky,
// 2x2 / 2 max pool (darknet L0 and the reweighting net's L0) -- in ONE sweep over its activation.
//
// The unfused sequence moves the largest activation of the network five times: fsd_bn_act_pool_bwd reads dz and y and
// writes dt, the per-channel sums are reduced, fsd_conv3x3_wgrad_c4_bnfused reads dt and y again.  A first layer needs
// no data gradient, so dt is only ever consumed by per-channel sums and by the weight gradient, and the BatchNorm
// backward is affine in dt:
//       dy = c1 (dt - c2 - xhat c3),     c2 = mean(dt),  c3 = mean(dt xhat),  c1 = gamma invstd     (fsd_bn_bwd_finalize)
//   =>  dW[co][col] = sum_p dy[p,co] x[p+tap,ci] = c1[co] ( S1[co][col] - c2[co] S2[col] - c3[co] S3[co][col] )
//       S1 = sum_p dt[p,co] x[p+tap,ci]      S2 = sum_p x[p+tap,ci]      S3 = sum_p xhat[p,co] x[p+tap,ci]
// S1, S2, S3 and the two BatchNorm sums (sum dt, sum dt xhat) are all accumulated by the pass that FORMS dt from
// (dz, y): dt is never written, y is read once.  A small second kernel folds the per-workgroup partials (fp64, fixed
// order) and applies the coefficients.  HBM traffic: dz (a quarter of the activation) + y + the 16-byte input pixels,
// against 2 x dz/dt + 2 x y + dt before (416x416x32 fp32 at B = 64: 1.9 GB instead of 6.0 GB).
//
// MFMA formulation (v_mfma_f32_32x32x2_f32, no LDS staging, as in wgrad_first_kernel): a wave owns a run of pooling
// cells and takes them in pairs; a pair is four k-steps (the four window pixels) of two pixels (one from each cell).
// Lane (c, h): A operand = channel c of the pixel of cell h, B operand = (tap, ci) column c of the same pixel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "fsdet.h"
#include "profile.hpp"
#include "ew_types.hpp"

namespace fsd_conv { bool f32_split_on(); }     // conv.hip: the arithmetic of the fp32 GEMMs (fsd_f32_gemm_mode)

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

// pairs of fp32 values -> packed bf16 planes v = v1 + v2 + v3 (round-to-nearest, exact residuals; one v_cvt_pk_bf16_f32 per
// pair and plane -- the idiom of conv.hip's in-kernel split).  PLANES = how many are wanted.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt2(float a, float b) {
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float lo_f(unsigned pk) { return __builtin_bit_cast(float, pk << 16); }
__device__ __forceinline__ float hi_f(unsigned pk) { return __builtin_bit_cast(float, pk & 0xffff0000u); }
template <int PLANES>
__device__ __forceinline__ void split_pair(float a, float b, unsigned& p1, unsigned& p2, unsigned& p3) {
  p1 = cvt2(a, b);
  p2 = p3 = 0u;
  if constexpr (PLANES >= 2) {
    const float ra = a - lo_f(p1), rb = b - hi_f(p1);
    p2 = cvt2(ra, rb);
    if constexpr (PLANES >= 3) p3 = cvt2(ra - lo_f(p2), rb - hi_f(p2));
  }
}
// eight values (the 4 + 4 window pixels of a lane's two cells) -> PLANES operand registers of v_mfma_f32_32x32x16_bf16
template <int PLANES>
__device__ __forceinline__ void split8(const float (&a)[4], const float (&b)[4], bf16x8_t& o1, bf16x8_t& o2, bf16x8_t& o3) {
  unsigned w1[4], w2[4], w3[4];
  split_pair<PLANES>(a[0], a[1], w1[0], w2[0], w3[0]);
  split_pair<PLANES>(a[2], a[3], w1[1], w2[1], w3[1]);
  split_pair<PLANES>(b[0], b[1], w1[2], w2[2], w3[2]);
  split_pair<PLANES>(b[2], b[3], w1[3], w2[3], w3[3]);
  o1 = __builtin_bit_cast(bf16x8_t, make_uint4(w1[0], w1[1], w1[2], w1[3]));
  o2 = __builtin_bit_cast(bf16x8_t, make_uint4(w2[0], w2[1], w2[2], w2[3]));
  o3 = __builtin_bit_cast(bf16x8_t, make_uint4(w3[0], w3[1], w3[2], w3[3]));
}
using fsd_ew::bf16_t;


struct FirstBwdArgs {
  const void* dz; const void* y;          // float or bf16 (template parameter); leading dimensions in ELEMENTS
  const float* scale; const float* shift; const float* mean; const float* invstd; const float* x;
  float* ws;                               // [blocks][2][Cout][36]: S1, S3
  float* ws2;                              // [blocks][36]: S2
  float* partial;                          // [blocks][Cout][2]: sum dt, sum dt*xhat
  unsigned dz_ld, y_ld, x_ld;
  int H, W, OH, OW, Cout;
  long long cells;
  int cpw;                                 // cells per wave (multiple of 8: a wave works on PAIRS of cells)
  float slope;
};

// SIDE = false: 3 input channels, the 27 (tap, ci) columns fit one 32-wide MFMA tile.
// SIDE = true : 4 input channels, taps 0..7 in the tile and the ninth tap as FMA side sums per lane.
//
// Round 6 lane mapping.  A wave walks its run of pooling cells two at a time: lane (c, h) owns the WHOLE 2x2 window of cell
// 2i + h for channel c (A role) and column c of the same four pixels (B role), so the pool winner is found in the lane (no
// lane exchange), only the winner carries a gradient (one select chain instead of four), and one MFMA pair per window pixel
// takes k = h from the two cells.  The B operand -- 27 (36) values around each pixel -- no longer comes from per-lane
// gathers: the 4 x 6-pixel input patch of the cell pair is fetched by ONE 16-byte load on 24 lanes (out-of-image pixels as
// zeros), parked in a wave-private LDS slot, and every lane picks its taps with ds_read at constant offsets.  Before: lane
// (c, h) held pixel COLUMN h of one cell, both lanes of a pair worked out the same winner through two lane exchanges, and
// five loads per cell -- two of them gathers over ~8 cache lines each -- kept the CU's one texture path busier than its
// four matrix pipes: 0.75 ms on the 416x416 layer in both storage types, 2.6x the MFMA time; now 0.57 / 0.56 ms (fp32 / bf16
// storage).  What bounds it now (tools/probes/first_bwd_ablate.sh, B = 64): with the S3 MFMAs compiled out 0.42 ms, with all
// of them out 0.39 ms fp32 (= the 1.95 GB of dz, y and x at 5 TB/s) / 0.26 ms bf16 -- the fp32-MFMA work (0.29 ms at the
// peak rate) and the HBM-bound sweep overlap only partly.  The workgroup count does not matter (1024..5120: +-2 %).
// SPLIT (fp32 storage): the products of both sums as the six bf16-MFMA terms of three-way split operands -- the arithmetic of
// the library's fp32 GEMMs (conv.hip; error against fp64 at or below the fp32 MFMA's) -- instead of v_mfma_f32_32x32x2_f32,
// which stays for fsd_f32_gemm_mode(0).  bf16 storage always runs on the bf16 MFMA.
template <bool SIDE, typename T, bool SPLIT>
__device__ __forceinline__ void first_bwd_body(const FirstBwdArgs& p) {
  constexpr unsigned ES = sizeof(T);
  __shared__ __attribute__((aligned(16))) float s_out[4][32 * 36 + 36 + 64];   // per wave: the input patch while sweeping;
                                                                                // afterwards a [32][36] tile, S2[36], BN sums [32][2]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int c = lane & 31, h = lane >> 5;
  const int co = blockIdx.y * 32 + c;
  const float sc = p.scale[co], sh = p.shift[co], mu = p.mean[co], is = p.invstd[co];
  const int tap = SIDE ? c >> 2 : c / 3, ci = SIDE ? c & 3 : c - 3 * (c / 3);
  const bool col_ok = SIDE || c < 27;
  const int ky = tap / 3, kx = tap - 3 * ky;
  const int dyo = col_ok ? ky - 1 : 0, dxo = col_ok ? kx - 1 : 0;      // (columns 27..31 of the 3-channel tile: the centre tap)
  const long long c_begin = ((long long)blockIdx.x * 4 + wave) * p.cpw;
  const long long c_end = c_begin + p.cpw < p.cells ? c_begin + p.cpw : p.cells;
  const int len = (int)(c_end - c_begin);                    // cells of this wave (may be <= 0)
  f32x16 acc1, acc3;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc1[r] = 0.f; acc3[r] = 0.f; }
  float a81[4] = {0.f, 0.f, 0.f, 0.f}, a83[4] = {0.f, 0.f, 0.f, 0.f}, s2c8[4] = {0.f, 0.f, 0.f, 0.f};
  float s1 = 0.f, s2 = 0.f, s2c = 0.f;
  if (len > 0) {
    const int OW = p.OW, OH = p.OH, W = p.W, H = p.H;
    // UNIFORM position of the pair's first cell: column, row within the image, global cell row (b * OH + cy), cell index
    int cx0 = (int)(c_begin % OW);
    const long long t0 = c_begin / OW;
    int cy0 = (int)(t0 % OH);
    unsigned r0 = (unsigned)t0, n0 = (unsigned)c_begin;
    int rel = 0;
    const char* dz_b = reinterpret_cast<const char*>(p.dz);
    const char* y_b = reinterpret_cast<const char*>(p.y);
    const char* x_b = reinterpret_cast<const char*>(p.x);
    // lane-constant byte offsets of the window pixels j = 2 * row + col from the pair's first pixel (the lane's cell is 2 h
    // pixels to the right)
    const unsigned l_dz = ((unsigned)h * p.dz_ld + (unsigned)co) * ES;
    unsigned l_y[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) l_y[j] = ((2u * h + (j & 1) + (unsigned)(j >> 1) * W) * p.y_ld + (unsigned)co) * ES;
    // the patch: image rows 2 cy0 - 1 .. + 2, columns 2 cx0 - 1 .. + 4, one 16-byte pixel per lane < 24
    const int prow = lane / 6, pcol = lane - 6 * prow;
    const bool p_lane = lane < 24;
    const int l_p = p_lane ? ((prow - 1) * W + (pcol - 1)) * (int)p.x_ld * 4 : 0;       // from the pair's first pixel, may be < 0
    float* patch = s_out[wave];
    // this lane's taps in the patch (float index): pixel j adds ((j >> 1) * 6 + (j & 1)) * 4
    const int l_b = ((1 + dyo) * 6 + (1 + 2 * h + dxo)) * 4 + ci;
    const int l_8 = (2 * 6 + (2 + 2 * h)) * 4;
    // 32-bit byte offsets everywhere (the launcher guarantees they fit)
    struct Group { float y[4]; float gz; f32x4 px; bool valid; bool in; bool live; };
    // Loads are unconditional from clamped (always mapped) offsets and the zero-selects happen at use, so the six loads of a
    // pair are in flight together, one pair ahead of the MFMAs (no branch in here: the compiler counts vmcnt exactly).
    auto load = [&](Group& g) {
      g.live = rel < len;
      const bool wrap = cx0 + 1 >= OW;                                     // the pair's second cell starts a new cell row:
      const bool valid = rel + h < len && !(wrap && h);                    // it is left to the next step
      const unsigned p00 = 2u * (n0 + r0 * (unsigned)OW);
      const unsigned o_safe = (unsigned)co * ES;                          // always mapped
      // (the offsets go through an empty asm: the compiler otherwise merges the loads from the safe offset, turns the selects
      // into divergent branches around them and waits with vmcnt(0) at every join)
      unsigned o_gz = valid ? n0 * p.dz_ld * ES + l_dz : o_safe;
      asm volatile("" : "+v"(o_gz));
      g.gz = fsd_ew::ld1<T>(reinterpret_cast<const T*>(dz_b + o_gz));
      const unsigned yq = p00 * p.y_ld * ES;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        unsigned o = valid ? yq + l_y[j] : o_safe;
        asm volatile("" : "+v"(o));
        g.y[j] = fsd_ew::ld1<T>(reinterpret_cast<const T*>(y_b + o));
      }
      const int yy = 2 * cy0 - 1 + prow, xx = 2 * cx0 - 1 + pcol;
      const bool in = g.live && p_lane && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
      unsigned o_p = in ? (unsigned)((int)(p00 * p.x_ld * 4u) + l_p) : 0u;
      asm volatile("" : "+v"(o_p));
      g.px = *reinterpret_cast<const f32x4*>(x_b + o_p);
      g.valid = valid;
      g.in = in;
      const int adv = wrap ? 1 : 2;
      rel += adv; n0 += adv; cx0 += adv;
      while (cx0 >= OW) {                 // next cell row
        cx0 -= OW; ++r0;
        if (++cy0 == OH) cy0 = 0;
      }
    };
    // everything of a pair but its MFMAs: patch through LDS, pool winner, BatchNorm algebra, the per-lane sums; hands back the
    // operands of the four k-steps: dq[j] (A of S1), xh[j] (A of S3), b[j] (B of both)
    auto prep = [&](const Group& g, float (&dqv)[4], float (&xh)[4], float (&b)[4]) {
      // hand the patch to the wave (wave-private LDS slot; the wave's own LDS operations execute in order)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (p_lane) *reinterpret_cast<f32x4*>(patch + 4 * lane) = g.in ? g.px : f32x4{0.f, 0.f, 0.f, 0.f};
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const bool valid = g.valid;
      f32x4 x8[SIDE ? 4 : 1];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v = patch[l_b + ((j >> 1) * 6 + (j & 1)) * 4];
        b[j] = valid ? v : 0.f;
        if constexpr (SIDE) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(patch + l_8 + ((j >> 1) * 6 + (j & 1)) * 4);
          x8[j] = valid ? w : f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      float a[4];
      bool pos[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float t = __builtin_fmaf(g.y[j], sc, sh);
        pos[j] = t > 0.f;
        a[j] = pos[j] ? t : t * p.slope;
        const float v = (g.y[j] - mu) * is;
        xh[j] = valid ? v : 0.f;
      }
      // the window in scan order, first maximum wins (like torch and fsd_bn_act_pool_bwd)
      const bool w1 = a[1] > a[0];
      const float m1 = w1 ? a[1] : a[0];
      const bool w2 = a[2] > m1;
      const float m2 = w2 ? a[2] : m1;
      const bool w3 = a[3] > m2;
      const bool win[4] = {!w1 && !w2 && !w3, w1 && !w2 && !w3, w2 && !w3, w3};
      const bool posw = (win[0] && pos[0]) || (win[1] && pos[1]) || (win[2] && pos[2]) || (win[3] && pos[3]);
      const float gz = valid ? g.gz : 0.f;
      const float dsel = posw ? gz : gz * p.slope;             // the one non-zero dt of the window
      const float xw = win[3] ? xh[3] : win[2] ? xh[2] : win[1] ? xh[1] : xh[0];
      s1 += dsel;
      s2 = __builtin_fmaf(dsel, xw, s2);
      // the weight gradient sees dt as the unfused path would have STORED it (bf16 mode: rounded)
      const float dq = fsd_ew::stored<T>(dsel);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        dqv[j] = win[j] ? dq : 0.f;
        s2c += b[j];
        if constexpr (SIDE) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            a81[i] = __builtin_fmaf(dqv[j], x8[j][i], a81[i]);
            a83[i] = __builtin_fmaf(xh[j], x8[j][i], a83[i]);
            s2c8[i] += x8[j][i];
          }
        }
      }
    };
    // fp32 storage: the products stay on the fp32 MFMA (one pair = four k-steps of two pixels)
    auto compute = [&](const Group& g) {
      float dqv[4], xh[4], b[4];
      prep(g, dqv, xh, b);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(dqv[j], b[j], acc1, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(xh[j], b[j], acc3, 0, 0, 0);
      }
    };
    // bf16 storage: dt is a bf16 value there, so S1 = dt x (x1 + x2 + x3) is three terms of v_mfma_f32_32x32x16_bf16 -- exact
    // products, fp32 accumulate -- and S3 = xhat x x takes the five terms down to 2^-16 of the three-way / two-way split
    // (xhat is formed from a bf16 activation: its own rounding is 2^-9).  TWO pairs fill the 16 k-slots of an instruction
    // (lane (c, h) brings the 4 + 4 window pixels of its two cells): 8 MFMAs of 32 cycles per four cells instead of 16 of 64.
    auto compute2 = [&](const Group& ga, const Group& gb) {
      float dqa[4], xha[4], ba[4], dqb[4], xhb[4], bb[4];
      prep(ga, dqa, xha, ba);
      prep(gb, dqb, xhb, bb);
      bf16x8_t b1, b2, b3, dq1, dq2, dq3, x1, x2, x3;
      split8<3>(ba, bb, b1, b2, b3);
      if constexpr (sizeof(T) == 2) {
        split8<1>(dqa, dqb, dq1, dq2, dq3);                      // (dq is a bf16 value: one plane holds it exactly)
        split8<2>(xha, xhb, x1, x2, x3);
      } else {
        split8<3>(dqa, dqb, dq1, dq2, dq3);
        split8<3>(xha, xhb, x1, x2, x3);
      }
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dq1, b1, acc1, 0, 0, 0);
      acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, b1, acc3, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dq1, b2, acc1, 0, 0, 0);
      acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, b2, acc3, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dq1, b3, acc1, 0, 0, 0);
      acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x2, b1, acc3, 0, 0, 0);
      acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x2, b2, acc3, 0, 0, 0);
      acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, b3, acc3, 0, 0, 0);
      if constexpr (sizeof(T) == 4) {                            // fp32 storage: the remaining terms of the six-term product
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dq2, b1, acc1, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x3, b1, acc3, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dq2, b2, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dq3, b1, acc1, 0, 0, 0);
      }
    };
    constexpr int kDepth = 4;
    Group g[kDepth];
    if constexpr (sizeof(T) == 2 || SPLIT) {
      load(g[0]);
      load(g[1]);
      while (g[0].live) {                 // (a pair past the end of the run loads from the safe offsets and adds zeros)
        load(g[2]);
        load(g[3]);
        __builtin_amdgcn_sched_barrier(0);
        compute2(g[0], g[1]);
        __builtin_amdgcn_sched_barrier(0);
        load(g[0]);
        load(g[1]);
        __builtin_amdgcn_sched_barrier(0);
        compute2(g[2], g[3]);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int d = 0; d < kDepth - 1; ++d) load(g[d]);
      while (g[0].live) {                 // (a pair past the end of the run loads from the safe offsets and adds zeros)
#pragma unroll
        for (int d = 0; d < kDepth; ++d) {
          load(g[(d + kDepth - 1) % kDepth]);
          __builtin_amdgcn_sched_barrier(0);
          compute(g[d]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    // the epilogue re-uses the patch slot
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  // ---- fold the four waves through LDS and write this workgroup's partials (fixed order: deterministic) ----
  float* so = s_out[wave];
  float* out = p.ws + ((size_t)blockIdx.x * 2 * p.Cout + (size_t)blockIdx.y * 32) * 36;
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    const f32x16& acc = which == 0 ? acc1 : acc3;
    float* a8 = which == 0 ? a81 : a83;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * h;        // output channel within the 32-tile
      if (col_ok) so[row * 36 + tap * 4 + ci] = acc[r];
    }
    if constexpr (SIDE) {
#pragma unroll
      for (int j = 0; j < 4; ++j) a8[j] += __shfl_xor(a8[j], 32, 64);
      if (h == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) so[c * 36 + 32 + j] = a8[j];
      }
    } else {
      if (c < 9) {                                             // unused ci = 3 columns: keep them defined
#pragma unroll
        for (int r = 0; r < 16; ++r) so[((r & 3) + 8 * (r >> 2) + 4 * h) * 36 + c * 4 + 3] = 0.f;
      }
    }
    if (which == 1) {
      // S2 per column and the BatchNorm sums per channel ride along in the tail of the wave's LDS slice
      const float v2 = s2c + __shfl_xor(s2c, 32, 64);
      const float b1 = s1 + __shfl_xor(s1, 32, 64), b2 = s2 + __shfl_xor(s2, 32, 64);
      if (h == 0) {
        if (col_ok) so[32 * 36 + tap * 4 + ci] = v2;
        if (!SIDE && c < 9) so[32 * 36 + c * 4 + 3] = 0.f;
        so[32 * 36 + 36 + 2 * c] = b1;
        so[32 * 36 + 36 + 2 * c + 1] = b2;
      }
      if constexpr (SIDE) {
        // every lane of a half saw the same x8 values (they depend on the pixel, not on c): lane 0 / lane 32 carry them
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v8 = s2c8[j] + __shfl_xor(s2c8[j], 32, 64);
          if (lane == 0) so[32 * 36 + 32 + j] = v8;
        }
      }
    }
    __syncthreads();
    float* o = out + (size_t)which * p.Cout * 36;
    for (int e = threadIdx.x; e < 32 * 36; e += 256) o[e] = s_out[0][e] + s_out[1][e] + s_out[2][e] + s_out[3][e];
    if (which == 1) {
      if (blockIdx.y == 0 && threadIdx.x < 36) {
        const int e = 32 * 36 + threadIdx.x;
        p.ws2[(size_t)blockIdx.x * 36 + threadIdx.x] = s_out[0][e] + s_out[1][e] + s_out[2][e] + s_out[3][e];
      }
      if (threadIdx.x >= 64 && threadIdx.x < 128) {
        const int e = 32 * 36 + 36 + (threadIdx.x - 64);             // (channel, which sum) pairs
        const int cc = (threadIdx.x - 64) >> 1, w = (threadIdx.x - 64) & 1;
        p.partial[((size_t)blockIdx.x * p.Cout + blockIdx.y * 32 + cc) * 2 + w] =
            s_out[0][e] + s_out[1][e] + s_out[2][e] + s_out[3][e];
      }
    }
    __syncthreads();
  }
}

template <bool SIDE>
__global__ __launch_bounds__(256) void first_bwd_kernel(FirstBwdArgs p) { first_bwd_body<SIDE, float, false>(p); }

// fp32 storage, split arithmetic (the default): 0.568 -> 0.51 ms on the 416x416 layer at B = 64 -- 12 MFMAs of 32 cycles per
// four cells instead of 16 of 64, paid for with ~50 more VALU instructions per cell pair for the three-way splits.
template <bool SIDE>
__global__ __launch_bounds__(256) void first_bwd_s_kernel(FirstBwdArgs p) { first_bwd_body<SIDE, float, true>(p); }

// (A branch to a select-free variant of the sweep for pairs that need no operand zeroed -- 97 % of them -- was measured: the
// compiler loses the counted vmcnt across it and spills, 0.40 -> 1.09 ms.  The selects stay.)
// bf16 storage.  Three waves per SIMD: left to itself the compiler takes 188 registers (two waves) and parks 32 values in
// accumulator registers, 96 moves per loop trip (0.465 ms); held to three waves it needs 157 and none (0.39 ms); four waves
// spill to scratch.  The fp32 kernel is slower with the same hint (0.57 -> 0.59 ms) and keeps the default.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void first_bwd_h_kernel(FirstBwdArgs p) {
  first_bwd_body<false, bf16_t, true>(p);
}
// (the 4-channel variant carries the ninth tap's side sums: held to three waves it spills 240 bytes to scratch -- the 80 x
// 416x416 supports of the COCO-shaped episode took 2 ms longer; it keeps the compiler's own allocation: 0.74 ms for those 80
// supports, 0.84 with fp32 storage, against 0.92 / 0.92 on the fp32 MFMA)
__global__ __launch_bounds__(256) void first_bwd_h4_kernel(FirstBwdArgs p) { first_bwd_body<true, bf16_t, true>(p); }

// dW[co][ci][ky][kx] = c1 (sum_b S1 - c2 sum_b S2 - c3 sum_b S3): one workgroup per (output channel, 9 of the 36 columns),
// 28 block lanes x 9 columns, fp64 sums in a fixed order.  (One workgroup per channel with 4 block lanes was a chain of
// 512 dependent-latency iterations on 32 workgroups: 141 us for 19 MB.)
constexpr int kFoldLanes = 28, kFoldCols = 9;
__global__ __launch_bounds__(256) void first_bwd_fold_kernel(const float* __restrict__ ws, const float* __restrict__ ws2,
                                                            const float* __restrict__ coef, float* __restrict__ dw,
                                                            int blocks, int cout, int cin) {
  __shared__ double s_acc[kFoldLanes][3][kFoldCols];
  const int co = blockIdx.x;
  const int cl = threadIdx.x % kFoldCols, ry = threadIdx.x / kFoldCols;
  const int col = blockIdx.y * kFoldCols + cl;
  if (ry < kFoldLanes) {
    double a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int b = ry; b < blocks; b += kFoldLanes) {
      a1 += (double)ws[(((size_t)b * 2 + 0) * cout + co) * 36 + col];
      a3 += (double)ws[(((size_t)b * 2 + 1) * cout + co) * 36 + col];
      a2 += (double)ws2[(size_t)b * 36 + col];
    }
    s_acc[ry][0][cl] = a1; s_acc[ry][1][cl] = a2; s_acc[ry][2][cl] = a3;
  }
  __syncthreads();
  if (ry == 0) {
    double a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int l = 0; l < kFoldLanes; ++l) { a1 += s_acc[l][0][cl]; a2 += s_acc[l][1][cl]; a3 += s_acc[l][2][cl]; }
    const double c1 = coef[co], c2 = coef[cout + co], c3 = coef[2 * cout + co];
    const int tap = col >> 2, ci = col & 3;
    if (ci < cin) dw[((size_t)co * cin + ci) * 9 + tap] = (float)(c1 * (a1 - c2 * a2 - c3 * a3));
  }
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

inline int fb_blocks(long long cells) {
  long long b = (cells + 4 * 64 - 1) / (4 * 64);              // at least 64 cells (256 pixels) per wave
  static const char* env = FSD_TUNE("FSD_FB_BLOCKS");         // (tuning aid)
  const long long cap = env ? atoll(env) : 2048;
  return (int)(b < 1 ? 1 : b > cap ? cap : b);
}

template <typename T>
int accum_impl(const T* dz, long long dz_ld, const T* y, long long y_ld, const float* scale, const float* shift,
               const float* mean, const float* invstd, float slope, const float* x, long long x_ld, void* workspace,
               size_t workspace_bytes, float* partial, int batch, int height, int width, int cin, int cout,
               hipStream_t stream) {
  (void)hipGetLastError();
  if (!dz || !y || !scale || !shift || !mean || !invstd || !x || !workspace || !partial || batch < 1 || height < 2 || width < 2)
    return FSD_ERR_ARG;
  if (cin < 1 || cin > 4 || cout % 32 || (height & 1) || (width & 1) || x_ld < 4 || (x_ld & 3) || dz_ld < cout || y_ld < cout)
    return FSD_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) & 15)) return FSD_ERR_ARG;
  const long long pixels = (long long)batch * height * width, cells = pixels / 4;
  const long long ld_max = dz_ld > y_ld ? (dz_ld > x_ld ? dz_ld : x_ld) : (y_ld > x_ld ? y_ld : x_ld);
  if ((pixels + 2LL * width + 18) * ld_max * 4 >= 0xffffffffLL) return FSD_ERR_UNSUPPORTED;      // 32-bit byte offsets
  const int blocks = fb_blocks(cells);
  if (workspace_bytes < fsd_first_layer_bwd_workspace_bytes(batch, height, width, cout)) return FSD_ERR_WORKSPACE;
  FirstBwdArgs a;
  a.dz = dz; a.y = y; a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd; a.x = x;
  a.ws = reinterpret_cast<float*>(workspace);
  a.ws2 = a.ws + (size_t)blocks * 2 * cout * 36;
  a.partial = partial;
  a.dz_ld = (unsigned)dz_ld; a.y_ld = (unsigned)y_ld; a.x_ld = (unsigned)x_ld;
  a.H = height; a.W = width; a.OH = height / 2; a.OW = width / 2; a.Cout = cout; a.cells = cells;
  a.cpw = round_up((int)((cells + (long long)blocks * 4 - 1) / ((long long)blocks * 4)), 8);
  a.slope = slope;
  fsd_prof::Scope prof(fsd_prof::kFirst, (double)pixels * (sizeof(T) * cout * 1.25 + 16.0), stream);
  const dim3 grid(blocks, cout / 32);
  if constexpr (sizeof(T) == 2) {
    if (cin == 4) FSD_LAUNCH(first_bwd_h4_kernel, grid, dim3(256), 0, stream, a);
    else FSD_LAUNCH(first_bwd_h_kernel, grid, dim3(256), 0, stream, a);
  } else if (fsd_conv::f32_split_on()) {
    if (cin == 4) FSD_LAUNCH((first_bwd_s_kernel<true>), grid, dim3(256), 0, stream, a);
    else FSD_LAUNCH((first_bwd_s_kernel<false>), grid, dim3(256), 0, stream, a);
  } else {
    if (cin == 4) FSD_LAUNCH((first_bwd_kernel<true>), grid, dim3(256), 0, stream, a);
    else FSD_LAUNCH((first_bwd_kernel<false>), grid, dim3(256), 0, stream, a);
  }
  return (int)hipGetLastError();
}

}  // namespace

extern "C" int fsd_first_layer_bwd_rows(int batch, int height, int width) {
  return fb_blocks((long long)batch * height * width / 4);
}

extern "C" size_t fsd_first_layer_bwd_workspace_bytes(int batch, int height, int width, int cout) {
  const size_t blocks = (size_t)fb_blocks((long long)batch * height * width / 4);
  return blocks * (2 * (size_t)cout * 36 + 36) * sizeof(float);
}

extern "C" int fsd_first_layer_bwd_accum(const float* dz, long long dz_ld, const float* y, long long y_ld,
                                         const float* scale, const float* shift, const float* mean, const float* invstd,
                                         float slope, const float* x, long long x_ld, void* workspace,
                                         size_t workspace_bytes, float* partial, int batch, int height, int width,
                                         int cin, int cout, hipStream_t stream) {
  return accum_impl<float>(dz, dz_ld, y, y_ld, scale, shift, mean, invstd, slope, x, x_ld, workspace, workspace_bytes,
                           partial, batch, height, width, cin, cout, stream);
}

extern "C" int fsd_first_layer_bwd_accum_h(const void* dz, long long dz_ld, const void* y, long long y_ld,
                                           const float* scale, const float* shift, const float* mean,
                                           const float* invstd, float slope, const float* x, long long x_ld,
                                           void* workspace, size_t workspace_bytes, float* partial, int batch, int height,
                                           int width, int cin, int cout, hipStream_t stream) {
  return accum_impl<bf16_t>(static_cast<const bf16_t*>(dz), dz_ld, static_cast<const bf16_t*>(y), y_ld, scale, shift, mean,
                            invstd, slope, x, x_ld, workspace, workspace_bytes, partial, batch, height, width, cin, cout,
                            stream);
}

extern "C" int fsd_first_layer_bwd_fold(const void* workspace, size_t workspace_bytes, const float* coef, float* dw_oihw,
                                        int batch, int height, int width, int cin, int cout, hipStream_t stream) {
  (void)hipGetLastError();
  if (!workspace || !coef || !dw_oihw || cin < 1 || cin > 4 || cout < 1) return FSD_ERR_ARG;
  if (workspace_bytes < fsd_first_layer_bwd_workspace_bytes(batch, height, width, cout)) return FSD_ERR_WORKSPACE;
  const int blocks = fb_blocks((long long)batch * height * width / 4);
  const float* ws = reinterpret_cast<const float*>(workspace);
  FSD_LAUNCH(first_bwd_fold_kernel, dim3(cout, 36 / kFoldCols), dim3(256), 0, stream, ws, ws + (size_t)blocks * 2 * cout * 36, coef,
                     dw_oihw, blocks, cout, cin);
  return (int)hipGetLastError();
}
