// bf16 storage mode: direct 3x3 convolution of the NARROW layers (32 -> 64 channels and its data gradient 64 -> 32: darknet
// L2 at 208x208, the reweighting net's second layer), activations and outputs bf16 NHWC, fp32 accumulate on
// v_mfma_f32_32x32x16_bf16.
//
// These layers are HBM-bound (L2 at B = 64: 177 MB in, 354 MB out, 102 GFLOP = 0.11 ms at 5 TB/s against 0.04 ms of
// matrix time) but ran at 0.245 ms (forward) / 0.320 ms (data gradient) on conv_bf16_dma_kernel: that kernel walks
// K = taps x Cin chunk by chunk and stages the 128 input rows of ONE tap per chunk, so every input pixel crosses
// L2 -> LDS nine times per workgroup, behind nine barriers, for 36 MFMAs per wave.  Here
//   * a workgroup is PERSISTENT (two per CU) and walks a contiguous run of 8 x 16 output-pixel blocks;
//   * the whole weight tensor (9 taps x Cin x Cout = 18432 bf16) lives in REGISTERS as MFMA B fragments (36 fragments =
//     144 VGPRs per lane), fetched once per workgroup from the packed weights of fsd_pack_conv_weight_bf16;
//   * per block the 10 x 18 halo patch is staged ONCE by LDS-DMA (16 bytes per lane, lane-linear in LDS; the piece a lane
//     fetches is XOR-permuted inside its pixel row so that the fragment reads are conflict-free), in a ring of buffers:
//     the patches of the next 4 (Cin = 32) / 2 (Cin = 64) blocks are in flight under the MFMAs and stores of a block,
//     behind a COUNTED vmcnt.  One barrier per block;
//   * the A fragment of tap (dy, dx) is one ds_read_b128 at a pixel offset of dy * 18 + dx in the patch; the MFMA row m of
//     a wave's 2 x 16 pixels is mapped to (row, column) so that each 16-lane service group of the read covers 16
//     CONSECUTIVE patch pixels (conv_halo.hip's pix_of);
//   * epilogue: accumulator columns are assigned channel 2l / 2l+1 (a free choice of which weight row a lane loads), one
//     v_cvt_pk_bf16_f32 per pair; 64 outputs: through a wave-private LDS tile and out as 16-byte stores (8 pixels x 128 B
//     per instruction); 32 outputs: lane pairs swap a value and store 4 bytes, 64 contiguous bytes per pixel.
//     BatchNorm partial sums stay in registers across the blocks of a workgroup and leave as ONE partial row per
//     workgroup (fsd_conv2d_h_partial_rows reports min(blocks, 512) rows for these shapes).
// The 64 -> 128 layers (darknet L4 / L6 at 104 x 104) run the same loop on 8 waves (2 pixel groups x 4 channel groups, one
// workgroup per CU): every wave still holds 144 registers of weights.
// Shapes: ksize 3, (Cin, Cout) = (32, 64), (64, 32) or (64, 128), H % 8 == 0, W % 16 == 0 -- for 64 -> 128 also W % 16 == 8 (the
// last block of a row is then half outside: W = 104; see halo_h_ok for why only there) -- bf16 NHWC output; everything else
// stays on conv_bf16_dma_kernel.  FSD_CONV_HALO=0 switches it off (the
// switch of the fp32 twin).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "fsdet.h"
#include "conv_common.hpp"
#include "profile.hpp"

namespace {

using namespace fsd_conv;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

constexpr int kBH = 8, kBW = 16;                   // output block
constexpr int kHW = kBW + 2, kHH = kBH + 2;        // halo patch 18 x 10
constexpr int kHaloPx = kHW * kHH;                 // 180 pixels

__device__ __attribute__((aligned(16))) u16 g_zero_page_hh[64];    // 128 zero bytes: source of the out-of-image pieces

struct HaloHArgs {
  const u16* x;          // bf16 NHWC, pixel stride x_ld elements
  const u16* w;          // packed bf16 weights [rows][Kpad], K = tap * Cin + ci (either packing mode)
  const float* bias;     // [Cout] or null
  u16* y;                // bf16 NHWC, pixel stride y_ld elements
  float* bn_partial;     // [gridDim.x][Cout][2] or null
  unsigned x_ld, y_ld;
  int H, W, Kpad;
  int bx, by;            // blocks per image row / column
  int blocks;            // B * bx * by
  float slope;
};

__device__ __forceinline__ void dma16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2(float lo, float hi) {      // one v_cvt_pk_bf16_f32 (round to nearest even)
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// MFMA row m (0..31) of a wave -> (row, column) of its 2 x 16 output pixels: the 16-lane service groups of a ds_read_b128
// ({0-3, 12-15, 20-27} / {4-11, 16-19, 28-31}, MI355X_MICROARCH.md) each take one pixel row.
__device__ __forceinline__ void pix_of(int m, int& py, int& px) {
  const bool g1 = (m >= 4 && m < 12) || (m >= 16 && m < 20) || m >= 28;
  py = g1 ? 1 : 0;
  px = m < 4 ? m : m < 12 ? m - 4 : m < 20 ? m - 8 : m < 28 ? m - 12 : m - 16;
}

// Block coordinates carried from block to block (no divisions in the loop).
struct BlockPos {
  int bxi, byi, img;
  __device__ __forceinline__ void init(int blk, int bx, int by) {
    bxi = blk % bx;
    const int t2 = blk / bx;
    byi = t2 % by;
    img = t2 / by;
  }
  __device__ __forceinline__ void next(int bx, int by) {
    if (++bxi == bx) {
      bxi = 0;
      if (++byi == by) { byi = 0; ++img; }
    }
  }
};

// 32-channel outputs: lane l holds channel l of accumulator rows (r, r + 1).  Both lanes of a pair pack their own two rows
// (one v_cvt_pk_bf16_f32), swap the packed word with the partner on the VALU (DPP quad_perm [1, 0, 3, 2]; __shfl_xor would
// go through the LDS crossbar) and pick two halves with ONE v_perm_b32 whose selector is a lane constant:
//   even lane: channels (l, l+1) of row r     = (own low half,  partner's low half)
//   odd lane:  channels (l-1, l) of row r + 1 = (partner's high half, own high half)
__device__ __forceinline__ unsigned pair_rows(float a, float b, unsigned sel) {
  const unsigned own = pack2(a, b);
  const unsigned other = (unsigned)__builtin_amdgcn_update_dpp((int)own, (int)own, 0xB1, 0xf, 0xf, false);
  return __builtin_amdgcn_perm(other, own, sel);
}

// CIN -> COUT on WP x WN waves: a wave owns TM = 4 / WP sub-tiles of 32 pixels (2 image rows x 16) and TN = COUT / 32 / WN
// accumulator columns; its weights are 9 x CIN/16 x TN register fragments (144 VGPRs in every configuration).
//   (32,  64, 4, 1): 4 waves, two workgroups per CU (darknet L2 forward)
//   (64,  32, 4, 1): its data gradient
//   (64, 128, 2, 4): 8 waves, ONE workgroup per CU (darknet L4 / L6 forward, 104 x 104)
// EPI: bias and leaky slope in the epilogue (the inference form); the training path (neither) compiles without them.
template <int CIN, int COUT, int WP, int WN, bool EPI>
__global__ __launch_bounds__(64 * WP * WN, (WP * WN == 4 ? 2 : 1)) void conv3x3_halo_h_kernel(HaloHArgs p) {
  constexpr int NT = 64 * WP * WN;
  constexpr int KS = CIN / 16;                      // k-steps per tap
  constexpr int TM = 4 / WP, TN = COUT / 32 / WN;
  static_assert(TM * WP == 4 && TN * WN * 32 == COUT && (TN == 1 || TN == 2) && 9 * KS * TN == 36, "144 weight registers");
  constexpr int ROWB = CIN * 2;                     // bytes of one patch pixel
  constexpr int PPR = ROWB / 16;                    // 16-byte pieces per pixel: 4 / 8
  constexpr int PIECES = kHaloPx * PPR;             // 720 / 1440
  constexpr int PASSES = (PIECES + NT - 1) / NT;
  constexpr int BUF = PASSES * NT * 16;             // bytes of one patch buffer (the tail pieces are zeros)
  // Slot q of patch pixel (hy, hx) holds the 16-byte piece q ^ ((hx >> SWS) & SWM).  A 16-lane service group of a fragment
  // read covers 16 consecutive columns of one patch row: (pixel address mod 256 B, slot) then takes 16 distinct values
  // (the patch is 18 wide and 18 * ROWB is a multiple of 256 B away from... the row offset only shifts the pixel part),
  // and because the permutation depends on the column alone, the fragment address of tap (ky, kx), sub-tile i is
  // [lane base of (kx, k-step)] + (ky + 2 i) * 18 * ROWB: an immediate offset, no address arithmetic per read.
  constexpr int SWS = CIN == 32 ? 2 : 1, SWM = PPR - 1;
  // D patches in flight ahead of the one being computed (NB = D + 1 buffers): the MFMAs of a block take ~2 k cycles, an
  // HBM fetch under load more, and a CU needs ~40 KB in flight to keep its share of 5 TB/s (Little).  The wait for a
  // patch is COUNTED: loads and stores leave the vmcnt queue in issue order, so everything but the (D - 1) * PASSES pieces
  // and D * STORES stores issued after it may still be in flight (the first D blocks of a run, with fewer stores behind
  // them, wait for the stores as well).
  constexpr int D = CIN == 32 ? 4 : COUT == 32 ? 2 : 3, NB = D + 1;
  // 16-byte stores through a wave-private LDS tile [32 TM pixels][32 TN channels]; the 64 -> 32 configuration (two
  // workgroups x three 24 KB patches fill the LDS) stores 4 bytes per lane straight from registers
  constexpr bool LDS_STORE = !(CIN == 64 && COUT == 32);
  constexpr int TILE_B = TM * TN * 2048;            // bytes of a wave's store tile
  constexpr int STORES = LDS_STORE ? TILE_B / 1024 : 8 * TM;      // global stores per lane and block
  constexpr int KEEP = (D - 1) * PASSES + D * STORES;
  static_assert(KEEP <= 63, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(16))) unsigned char s_patch_all[];      // [NB][BUF] (+ the waves' store tiles)
  __shared__ float s_stat[WP][COUT][2];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pg = wave / WN, ng = wave - pg * WN;    // pixel group (TM sub-tiles) / channel group (32 TN channels)
  const int c_lane = lane & 31, fh = lane >> 5;

  // ---- this workgroup's run of blocks ----
  const int G = gridDim.x, g = blockIdx.x;
  const int b_begin = (int)((long long)p.blocks * g / G), b_end = (int)((long long)p.blocks * (g + 1) / G);

  // ---- weights -> registers: fragment (tap, ks, j) = 8 consecutive input channels of output channel ch(j) ----
  bf16x8 wf[9][KS][TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int ch = ng * 32 * TN + (TN == 2 ? 2 * c_lane + j : c_lane);
    const u16* wr = p.w + (long long)ch * p.Kpad + 8 * fh;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) wf[t][ks][j] = *reinterpret_cast<const bf16x8*>(wr + t * CIN + ks * 16);
  }
  float bv[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
    bv[j] = (EPI && p.bias != nullptr) ? p.bias[ng * 32 * TN + (TN == 2 ? 2 * c_lane + j : c_lane)] : 0.f;

  // ---- staging roles: piece e = tid + NT i of the patch = (pixel e / PPR, slot e % PPR) ----
  int s_rel[PASSES];                                // element offset of the piece from the block's first pixel (may be < 0)
  unsigned s_hyx[PASSES];                           // (hy << 8) | hx, or 0xffff for the pieces past the patch
#pragma unroll
  for (int i = 0; i < PASSES; ++i) {
    const int e = tid + NT * i, hp = e / PPR, q = e - hp * PPR;
    const int hy = hp / kHW, hx = hp - hy * kHW;
    const int piece = q ^ ((hx >> SWS) & SWM);      // swizzle by the patch COLUMN (see the fragment addresses)
    s_rel[i] = ((hy - 1) * p.W + (hx - 1)) * (int)p.x_ld + piece * 8;
    s_hyx[i] = hp < kHaloPx ? (unsigned)((hy << 8) | hx) : 0xffffu;
  }
  const u16* zero_src = g_zero_page_hh;
  asm volatile("" : "+v"(zero_src));                // opaque: one address, not an s_getpc + s_load per use
  // One patch.  Interior blocks (no halo pixel outside the image: 78 % of them at 208 x 208) take the path without
  // per-piece predicates; blocks past the end of the run are staged as zeros (the counted wait needs the same number of
  // pieces every time).
  auto stage = [&](const BlockPos& bp, bool live, unsigned char* buf) {
    const int y0 = bp.byi * kBH, x0 = bp.bxi * kBW;
    const u16* base = p.x + (long long)((bp.img * p.H + y0) * p.W + x0) * p.x_ld;
    const bool interior = live && bp.bxi > 0 && x0 + kBW < p.W && bp.byi > 0 && bp.byi + 1 < p.by;
    if (interior) {
#pragma unroll
      for (int i = 0; i < PASSES; ++i) {
        int rel = s_rel[i];
        asm volatile("" : "+v"(rel));               // (p.x + rel is not to be kept as 64-bit lane pointers across the loop)
        const bool ok = i + 1 < PASSES || s_hyx[i] != 0xffffu;      // only the last pass has pieces past the patch
        dma16(ok ? base + rel : zero_src, buf + (NT * i + wave * 64) * 16);
      }
    } else {
#pragma unroll
      for (int i = 0; i < PASSES; ++i) {
        int rel = s_rel[i];
        asm volatile("" : "+v"(rel));
        const int hy = (int)(s_hyx[i] >> 8), hx = (int)(s_hyx[i] & 255u);
        const bool ok = live && s_hyx[i] != 0xffffu && (unsigned)(y0 - 1 + hy) < (unsigned)p.H &&
                        (unsigned)(x0 - 1 + hx) < (unsigned)p.W;
        dma16(ok ? base + rel : zero_src, buf + (NT * i + wave * 64) * 16);
      }
    }
  };

  // ---- fragment geometry ----
  // MFMA row m of a sub-tile -> pixel (py, px) of its 2 x 16 (pix_of).  For the accumulator rows of a lane, m = (r & 3) +
  // 8 (r >> 2) + 4 fh, this is px = r and py = fh ^ (1 for r in 4..11, else 0).
  int fpy, fpx;
  pix_of(c_lane, fpy, fpx);
  const int hr0 = (2 * TM * pg + fpy) * kHW + fpx;  // patch pixel of tap (0, 0) for this lane's pixel of sub-tile 0

  // BatchNorm sums of this lane's channel(s), two interleaved partial sums each: the packed adds / FMAs then take the
  // accumulator registers (r, r + 1) as they lie (pairing acc[0][r] with acc[1][r] cost 70 register moves per block)
  f32x2 st_s[TN], st_q[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) { st_s[j] = f32x2{0.f, 0.f}; st_q[j] = f32x2{0.f, 0.f}; }

  if (b_begin < b_end) {
    BlockPos cp, fp;                                // block being computed / block being fetched
    cp.init(b_begin, p.bx, p.by);
    fp = cp;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      stage(fp, b_begin + d < b_end, s_patch_all + d * BUF);
      fp.next(p.bx, p.by);
    }
    int cur = 0, fill = D;                          // buffer of the block being computed / of the patch issued this iteration
    for (int blk = b_begin; blk < b_end; ++blk) {
      if (blk - b_begin < D) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * PASSES) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP) : "memory");
      __syncthreads();                              // this block's patch has landed; the buffer of the previous block is free
      stage(fp, blk + D < b_end, s_patch_all + fill * BUF);
      fp.next(p.bx, p.by);
      const unsigned char* sA = s_patch_all + cur * BUF;
      // re-derived per block on purpose: hoisted out of the loop, the fragment addresses cost 18+ registers next to the
      // 144 of the weights (and spill them)
      int hr_b = hr0, fh_b = fh, fpx_b = fpx;
      asm volatile("" : "+v"(hr_b), "+v"(fh_b), "+v"(fpx_b));
      const unsigned char* fa[3][KS];               // lane base of (kx, k-step): patch pixel hr_b + kx, slot (2 ks + fh) ^ swizzle
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int sw = ((fpx_b + kx) >> SWS) & SWM;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) fa[kx][ks] = sA + (hr_b + kx) * ROWB + (((2 * ks + fh_b) ^ sw) << 4);
      }
      f32x16 acc[TM][TN];
      const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      // a group of taps at a time -- a kernel row (TM = 1) or a single tap (TM = 2): the scheduler otherwise hoists every
      // fragment read to the top, 72+ registers on top of the 144 of the weights
      constexpr int GT = TM == 1 ? 3 : 1;
#pragma unroll
      for (int t0 = 0; t0 < 9; t0 += GT) {
        bf16x8 af[GT][KS][TM];
#pragma unroll
        for (int tt = 0; tt < GT; ++tt) {
          const int t = t0 + tt;
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
              af[tt][ks][i] = *reinterpret_cast<const bf16x8*>(fa[t % 3][ks] + (t / 3 + 2 * i) * kHW * ROWB);
        }
#pragma unroll
        for (int tt = 0; tt < GT; ++tt)
#pragma unroll
          for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tt][ks][i], wf[t0 + tt][ks][j],
                                                                  (t0 | tt | ks) == 0 ? zero16 : acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- epilogue of the block ----
      const int x0 = cp.bxi * kBW;
      const bool ragged = x0 + kBW > p.W;           // last block of a row when W % 16 == 8: pixel columns 8..15 are outside
      u16* yb = p.y + (long long)((cp.img * p.H + cp.byi * kBH + 2 * TM * pg) * p.W + x0) * p.y_ld + ng * 32 * TN;
      cp.next(p.bx, p.by);
      if (ragged) {
        asm volatile("" ::: "memory");              // (a real branch: if-converted, this is 32 selects in EVERY block)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 8; r < 16; ++r) acc[i][j][r] = 0.f;      // px = r: keeps them out of the BatchNorm sums
      }
      if (p.bn_partial != nullptr) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
              const f32x2 v = {acc[i][j][r], acc[i][j][r + 1]};
              st_s[j] += v;
              st_q[j] = __builtin_elementwise_fma(v, v, st_q[j]);
            }
      }
      if constexpr (LDS_STORE) {
        unsigned char* tile = s_patch_all + NB * BUF + wave * TILE_B;
        if constexpr (TN == 2) {
          // [32 TM pixels][64 channels], 128-byte rows: lane l holds channels (2l, 2l+1)
          unsigned char* tl0 = tile + fh_b * 2048 + 4 * c_lane;            // rows 0..3, 12..15 of this lane: py = fh
          unsigned char* tl1 = tile + (fh_b ^ 1) * 2048 + 4 * c_lane;      // rows 4..11: py = fh ^ 1
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              float v0 = acc[i][0][r], v1 = acc[i][1][r];
              if (EPI) {
                v0 += bv[0]; v1 += bv[1];
                v0 = v0 > 0.f ? v0 : v0 * p.slope; v1 = v1 > 0.f ? v1 : v1 * p.slope;
              }
              *reinterpret_cast<unsigned*>(((r >= 4 && r < 12) ? tl1 : tl0) + i * 4096 + r * 128) = pack2(v0, v1);
            }
          // piece pc = lane + 64 t of the tile: pixel row t >> 1, column (lane >> 3) + 8 (t & 1), channels 8 (lane & 7) ..
          u16* yl = yb + (unsigned)(lane >> 3) * p.y_ld + (lane & 7) * 8;
#pragma unroll
          for (int t = 0; t < 4 * TM; ++t) {
            if (ragged && (t & 1)) continue;
            const uint4 v = *reinterpret_cast<const uint4*>(tile + (lane + 64 * t) * 16);
            *reinterpret_cast<uint4*>(yl + (unsigned)((t >> 1) * p.W + 8 * (t & 1)) * p.y_ld) = v;
          }
        } else {
          // [32 TM pixels][32 channels], 64-byte rows: lane l holds channel l; lane pairs swap one value per register pair,
          // the even lane then writes channels (l, l+1) of row r, the odd lane channels (l-1, l) of row r+1
          const bool odd = lane & 1;
          const unsigned sel = odd ? 0x03020706u : 0x05040100u;
          unsigned char* tl0 = tile + fh_b * 1024 + (odd ? 2 * (c_lane - 1) + 64 : 2 * c_lane);     // the odd lane writes row r + 1
          unsigned char* tl1 = tile + (fh_b ^ 1) * 1024 + (odd ? 2 * (c_lane - 1) + 64 : 2 * c_lane);
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
              float a = acc[i][0][r], b = acc[i][0][r + 1];
              if (EPI) {
                a += bv[0]; b += bv[0];
                a = a > 0.f ? a : a * p.slope; b = b > 0.f ? b : b * p.slope;
              }
              *reinterpret_cast<unsigned*>(((r >= 4 && r < 12) ? tl1 : tl0) + i * 2048 + r * 64) = pair_rows(a, b, sel);
            }
          // piece pc = lane + 64 t: pixel row t, column lane >> 2, channels 8 (lane & 3) ..
          u16* yl = yb + (unsigned)(lane >> 2) * p.y_ld + (lane & 3) * 8;
          const bool px_ok = !ragged || (lane >> 2) < 8;
#pragma unroll
          for (int t = 0; t < 2 * TM; ++t) {
            const uint4 v = *reinterpret_cast<const uint4*>(tile + (lane + 64 * t) * 16);
            if (px_ok) *reinterpret_cast<uint4*>(yl + (unsigned)(t * p.W) * p.y_ld) = v;
          }
        }
      } else {
        // 4-byte stores from registers (TN = 1): the same lane-pair exchange, 64 contiguous bytes per pixel
        const bool odd = lane & 1;
        const unsigned sel = odd ? 0x03020706u : 0x05040100u;
        u16* y0p = yb + (unsigned)(fh_b * p.W + (odd ? 1 : 0)) * p.y_ld + (odd ? c_lane - 1 : c_lane);          // py = fh
        u16* y1p = yb + (unsigned)((fh_b ^ 1) * p.W + (odd ? 1 : 0)) * p.y_ld + (odd ? c_lane - 1 : c_lane);    // py = fh ^ 1
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            if (ragged && r >= 8) continue;
            float a = acc[i][0][r], b = acc[i][0][r + 1];
            if (EPI) {
              a += bv[0]; b += bv[0];
              a = a > 0.f ? a : a * p.slope; b = b > 0.f ? b : b * p.slope;
            }
            u16* dst = ((r >= 4 && r < 12) ? y1p : y0p) + (unsigned)(2 * i * p.W + r) * p.y_ld;
            *reinterpret_cast<unsigned*>(dst) = pair_rows(a, b, sel);
          }
      }
      cur = cur + 1 == NB ? 0 : cur + 1;
      fill = fill + 1 == NB ? 0 : fill + 1;
    }
  }
  if (p.bn_partial != nullptr) {
    // one partial row per workgroup: lane halves, then the WP pixel-group waves of a channel group (fixed order)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float s = st_s[j][0] + st_s[j][1], q = st_q[j][0] + st_q[j][1];
      s += __shfl_xor(s, 32, 64);
      q += __shfl_xor(q, 32, 64);
      if (lane < 32) {
        const int ch = ng * 32 * TN + (TN == 2 ? 2 * c_lane + j : c_lane);
        s_stat[pg][ch][0] = s;
        s_stat[pg][ch][1] = q;
      }
    }
    __syncthreads();
    if (tid < COUT) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < WP; ++w) {
        s += s_stat[w][tid][0];
        q += s_stat[w][tid][1];
      }
      float* dst = p.bn_partial + ((long long)g * COUT + tid) * 2;
      dst[0] = s;
      dst[1] = q;
    }
  }
}

template <int CIN, int COUT, int WP, int WN, bool EPI>
int launch_halo_h(const HaloHArgs& a, int wgs, int lds, hipStream_t stream) {
  auto k = conv3x3_halo_h_kernel<CIN, COUT, WP, WN, EPI>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) return (int)e;
  FSD_LAUNCH(k, dim3(wgs), dim3(64 * WP * WN), lds, stream, a);
  return (int)hipGetLastError();
}

}  // namespace

bool fsd_conv::halo_h_ok(int height, int width, int cin, int cout, int ksize) {
  static const char* env = getenv("FSD_CONV_HALO");
  if (env && env[0] == '0') return false;
  if (!(ksize == 3 && ((cin == 32 && cout == 64) || (cin == 64 && cout == 32) || (cin == 64 && cout == 128)))) return false;
  // A ragged last block (W % 16 == 8) is taken by the 64 -> 128 configuration only: its epilogue issues every store under a
  // per-lane predicate, so the store count behind the COUNTED vmcnt wait is the same for every block.  The 32 -> 64 and
  // 64 -> 32 epilogues skip whole (wave-uniform) store instructions on a ragged block -- with more than D blocks per
  // workgroup the wait for a patch could then be satisfied while its own LDS-DMA pieces are still in flight -- so they take
  // whole blocks only (every stock size: these layers sit at S / 2, a multiple of 16); other widths stay on the GEMM kernel.
  return height % kBH == 0 && width % ((cin == 64 && cout == 128) ? 8 : kBW) == 0;
}

// strides / alignments the kernel's 16-byte DMA pieces and stores need (the plan queries and the launch agree on this)
bool fsd_conv::halo_h_layout_ok(const void* x, long long x_ld, const void* y, long long y_ld, int batch, int height, int width,
                                int cin, int cout) {
  const long long pixels = (long long)batch * height * width;
  if ((y_ld & 1) || (reinterpret_cast<uintptr_t>(y) & 3) || (x_ld & 7) || (reinterpret_cast<uintptr_t>(x) & 15) ||
      pixels >= 0x7fffffffLL || (width + 2) * x_ld >= 0x7fffffffLL || (4LL * width + 16) * y_ld >= 0x7fffffffLL)
    return false;
  if (!(cin == 64 && cout == 32) && ((y_ld & 7) || (reinterpret_cast<uintptr_t>(y) & 15))) return false;      // 16-byte stores
  return true;
}

// one BatchNorm partial row per persistent workgroup: two 4-wave workgroups per CU, or one 8-wave workgroup (64 -> 128)
int fsd_conv::halo_h_rows(int batch, int height, int width, int cin, int cout) {
  const long long blocks = (long long)batch * (height / kBH) * ((width + kBW - 1) / kBW);
  const int wgs = (cin == 64 && cout == 128) ? 256 : 512;
  return (int)(blocks < wgs ? blocks : wgs);
}

int fsd_conv::conv3x3_halo_h(const void* x, long long x_ld, const void* w_packed, int kpad, const float* bias, void* y,
                             long long y_ld, float* bn_partial, int batch, int height, int width, int cin, int cout,
                             float slope, hipStream_t stream) {
  const long long pixels = (long long)batch * height * width;
  if (!halo_h_layout_ok(x, x_ld, y, y_ld, batch, height, width, cin, cout)) return FSD_ERR_UNSUPPORTED;
  HaloHArgs a;
  a.x = static_cast<const u16*>(x); a.w = static_cast<const u16*>(w_packed); a.bias = bias; a.y = static_cast<u16*>(y);
  a.bn_partial = bn_partial; a.x_ld = (unsigned)x_ld; a.y_ld = (unsigned)y_ld;
  a.H = height; a.W = width; a.Kpad = kpad;
  a.bx = (width + kBW - 1) / kBW; a.by = height / kBH;
  a.blocks = batch * a.bx * a.by;
  a.slope = slope;
  const int wgs = halo_h_rows(batch, height, width, cin, cout);
  fsd_prof::Scope prof(fsd_prof::kGemmBf16, 2.0 * (double)pixels * cout * 9.0 * cin, stream);
  // dynamic LDS: (D + 1) patch buffers of PASSES * NT * 16 bytes + the waves' store tiles (see the kernel)
  const bool epi = bias != nullptr || slope != 1.f;
  if (cin == 64 && cout == 32) {
    const int lds = 3 * 6 * 4096;
    return epi ? launch_halo_h<64, 32, 4, 1, true>(a, wgs, lds, stream) : launch_halo_h<64, 32, 4, 1, false>(a, wgs, lds, stream);
  }
  if (cin == 32) {
    const int lds = 5 * 3 * 4096 + 4 * 4096;
    return epi ? launch_halo_h<32, 64, 4, 1, true>(a, wgs, lds, stream) : launch_halo_h<32, 64, 4, 1, false>(a, wgs, lds, stream);
  }
  const int lds = 4 * 3 * 8192 + 8 * 4096;
  return epi ? launch_halo_h<64, 128, 2, 4, true>(a, wgs, lds, stream) : launch_halo_h<64, 128, 2, 4, false>(a, wgs, lds, stream);
}
