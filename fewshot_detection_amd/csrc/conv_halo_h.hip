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
// Shapes: ksize 3, (Cin, Cout) = (32, 64) or (64, 32), H % 8 == 0, W % 16 == 0, bf16 NHWC output; everything else stays on
// conv_bf16_dma_kernel.  FSD_CONV_HALO=0 switches it off (the switch of the fp32 twin).
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
constexpr int kMaxWgs = 512;                       // two workgroups per CU

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

// EPI: bias and leaky slope in the epilogue (the inference form); the training path (neither) compiles without them.
template <int CIN, int COUT, bool EPI>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_h_kernel(HaloHArgs p) {
  static_assert((CIN == 32 && COUT == 64) || (CIN == 64 && COUT == 32), "32 -> 64 and its data gradient");
  constexpr int KS = CIN / 16;                      // k-steps per tap
  constexpr int TN = COUT / 32;
  constexpr int ROWB = CIN * 2;                     // bytes of one patch pixel
  constexpr int PPR = ROWB / 16;                    // 16-byte pieces per pixel: 4 / 8
  constexpr int PIECES = kHaloPx * PPR;             // 720 / 1440
  constexpr int PASSES = (PIECES + 255) / 256;      // 3 / 6
  constexpr int BUF = PASSES * 256 * 16;            // bytes of one patch buffer (the tail pieces are zeros)
  constexpr int SWS = CIN == 32 ? 2 : 1, SWM = PPR - 1;      // piece ^= (pixel >> SWS) & SWM
  // D patches in flight ahead of the one being computed (NB = D + 1 buffers): the MFMAs of a block take ~2 k cycles, an
  // HBM fetch under load more, and a CU needs ~40 KB in flight to keep its share of 5 TB/s (Little).  The wait for a
  // patch is COUNTED: loads and stores leave the vmcnt queue in issue order, so everything but the (D - 1) * PASSES pieces
  // and D * STORES stores issued after it may still be in flight (the first D blocks of a run, with fewer stores behind
  // them, wait for the stores as well).
  constexpr int D = CIN == 32 ? 4 : 2, NB = D + 1;
  constexpr int STORES = TN == 2 ? 4 : 8;           // global stores per lane and block
  constexpr int KEEP = (D - 1) * PASSES + D * STORES;
  static_assert(KEEP <= 63, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(16))) unsigned char s_patch_all[];      // [NB][BUF] (+ 4 x 4 KB store tiles, TN = 2)
  __shared__ float s_stat[4][COUT][2];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c_lane = lane & 31, fh = lane >> 5;

  // ---- this workgroup's run of blocks ----
  const int G = gridDim.x, g = blockIdx.x;
  const int b_begin = (int)((long long)p.blocks * g / G), b_end = (int)((long long)p.blocks * (g + 1) / G);

  // ---- weights -> registers: fragment (tap, ks, j) = 8 consecutive input channels of output channel ch(j) ----
  bf16x8 wf[9][KS][TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int ch = TN == 2 ? 2 * c_lane + j : c_lane;
    const u16* wr = p.w + (long long)ch * p.Kpad + 8 * fh;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) wf[t][ks][j] = *reinterpret_cast<const bf16x8*>(wr + t * CIN + ks * 16);
  }
  float bv[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) bv[j] = (EPI && p.bias != nullptr) ? p.bias[TN == 2 ? 2 * c_lane + j : c_lane] : 0.f;

  // ---- staging roles: piece e = tid + 256 i of the patch = (pixel e / PPR, slot e % PPR) ----
  int s_rel[PASSES];                                // element offset of the piece from the block's first pixel (may be < 0)
  unsigned s_hyx[PASSES];                           // (hy << 8) | hx, or 0xffff for the pieces past the patch
#pragma unroll
  for (int i = 0; i < PASSES; ++i) {
    const int e = tid + 256 * i, hp = e / PPR, q = e - hp * PPR;
    const int hy = hp / kHW, hx = hp - hy * kHW;
    const int piece = q ^ ((hp >> SWS) & SWM);
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
    const bool interior = live && bp.bxi > 0 && bp.bxi + 1 < p.bx && bp.byi > 0 && bp.byi + 1 < p.by;
    if (interior) {
#pragma unroll
      for (int i = 0; i < PASSES; ++i) {
        int rel = s_rel[i];
        asm volatile("" : "+v"(rel));               // (p.x + rel is not to be kept as 64-bit lane pointers across the loop)
        const bool ok = i + 1 < PASSES || s_hyx[i] != 0xffffu;      // only the last pass has pieces past the patch
        dma16(ok ? base + rel : zero_src, buf + (256 * i + wave * 64) * 16);
      }
    } else {
#pragma unroll
      for (int i = 0; i < PASSES; ++i) {
        int rel = s_rel[i];
        asm volatile("" : "+v"(rel));
        const int hy = (int)(s_hyx[i] >> 8), hx = (int)(s_hyx[i] & 255u);
        const bool ok = live && s_hyx[i] != 0xffffu && (unsigned)(y0 - 1 + hy) < (unsigned)p.H &&
                        (unsigned)(x0 - 1 + hx) < (unsigned)p.W;
        dma16(ok ? base + rel : zero_src, buf + (256 * i + wave * 64) * 16);
      }
    }
  };

  // ---- fragment geometry ----
  // MFMA row m of a wave -> pixel (py, px) of its 2 x 16 (pix_of).  For the accumulator rows of a lane, m = (r & 3) +
  // 8 (r >> 2) + 4 fh, this is px = r and py = fh ^ (1 for r in 4..11, else 0).
  int fpy, fpx;
  pix_of(c_lane, fpy, fpx);
  const int hr0 = (2 * wave + fpy) * kHW + fpx;     // patch pixel of tap (0, 0) for this lane's output pixel

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
      // re-derived per block on purpose: hoisted out of the loop, the 9 x KS fragment addresses cost 18+ registers next
      // to the 144 of the weights (and spill them)
      int hr_b = hr0, fh_b = fh;
      asm volatile("" : "+v"(hr_b), "+v"(fh_b));
      f32x16 acc[TN];
      const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      // one kernel row at a time (the scheduler otherwise hoists all 18 / 36 fragment reads to the top: 72+ registers on
      // top of the 144 of the weights)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        bf16x8 af[3][KS];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int hr = hr_b + ky * kHW + kx;
          const int sw = (hr >> SWS) & SWM;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks)
            af[kx][ks] = *reinterpret_cast<const bf16x8*>(sA + hr * ROWB + (((2 * ks + fh_b) ^ sw) << 4));
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kx][ks], wf[ky * 3 + kx][ks][j],
                                                               (ky | kx | ks) == 0 ? zero16 : acc[j], 0, 0, 0);      // first: C = 0 inline
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- epilogue of the block ----
      u16* yb = p.y + (long long)((cp.img * p.H + cp.byi * kBH + 2 * wave) * p.W + cp.bxi * kBW) * p.y_ld;
      cp.next(p.bx, p.by);
      if (p.bn_partial != nullptr) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const f32x2 v = {acc[j][r], acc[j][r + 1]};
            st_s[j] += v;
            st_q[j] = __builtin_elementwise_fma(v, v, st_q[j]);
          }
      }
      if constexpr (TN == 2) {
        // 16-byte stores through a wave-private 4 KB LDS tile [32 pixels][64 channels]: 4 store instructions per wave and
        // block instead of 16 four-byte ones
        unsigned char* tile = s_patch_all + NB * BUF + wave * 4096;
        unsigned char* t0 = tile + fh_b * 2048 + 4 * c_lane;              // rows 0..3, 12..15 of this lane: py = fh
        unsigned char* t1 = tile + (fh_b ^ 1) * 2048 + 4 * c_lane;        // rows 4..11: py = fh ^ 1
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v0 = acc[0][r], v1 = acc[1][r];
          if (EPI) {
            v0 += bv[0]; v1 += bv[1];
            v0 = v0 > 0.f ? v0 : v0 * p.slope; v1 = v1 > 0.f ? v1 : v1 * p.slope;
          }
          *reinterpret_cast<unsigned*>(((r >= 4 && r < 12) ? t1 : t0) + r * 128) = pack2(v0, v1);
        }
        // piece pc = lane + 64 i of the tile: pixel (i >> 1, (lane >> 3) + 8 (i & 1)), channels 8 (lane & 7) ..
        u16* yl = yb + (unsigned)(lane >> 3) * p.y_ld + (lane & 7) * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint4 v = *reinterpret_cast<const uint4*>(tile + (lane + 64 * i) * 16);
          *reinterpret_cast<uint4*>(yl + (unsigned)((i >> 1) * p.W + 8 * (i & 1)) * p.y_ld) = v;
        }
      } else {
        // 32 output channels: lane l holds channel l; lane pairs swap one value per register pair so that the even lane
        // stores channels (l, l+1) of row r and the odd lane channels (l-1, l) of row r+1: 4-byte stores
        const bool odd = lane & 1;
        u16* y0p = yb + (unsigned)(fh_b * p.W) * p.y_ld + (odd ? c_lane - 1 : c_lane);          // py = fh
        u16* y1p = yb + (unsigned)((fh_b ^ 1) * p.W) * p.y_ld + (odd ? c_lane - 1 : c_lane);    // py = fh ^ 1
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          float a = acc[0][r], b = acc[0][r + 1];
          if (EPI) {
            a += bv[0]; b += bv[0];
            a = a > 0.f ? a : a * p.slope; b = b > 0.f ? b : b * p.slope;
          }
          const float got = __shfl_xor(odd ? a : b, 1, 64);
          u16* dst = ((r >= 4 && r < 12) ? y1p : y0p) + (unsigned)(odd ? r + 1 : r) * p.y_ld;
          *reinterpret_cast<unsigned*>(dst) = odd ? pack2(got, b) : pack2(a, got);
        }
      }
      cur = cur + 1 == NB ? 0 : cur + 1;
      fill = fill + 1 == NB ? 0 : fill + 1;
    }
  }
  if (p.bn_partial != nullptr) {
    // one partial row per workgroup: lane halves, then the four waves (fixed order)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float s = st_s[j][0] + st_s[j][1], q = st_q[j][0] + st_q[j][1];
      s += __shfl_xor(s, 32, 64);
      q += __shfl_xor(q, 32, 64);
      if (lane < 32) {
        const int ch = TN == 2 ? 2 * c_lane + j : c_lane;
        s_stat[wave][ch][0] = s;
        s_stat[wave][ch][1] = q;
      }
    }
    __syncthreads();
    if (tid < COUT) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        s += s_stat[w][tid][0];
        q += s_stat[w][tid][1];
      }
      float* dst = p.bn_partial + ((long long)g * COUT + tid) * 2;
      dst[0] = s;
      dst[1] = q;
    }
  }
}

template <int CIN, int COUT, bool EPI>
int launch_halo_h(const HaloHArgs& a, int wgs, int lds, hipStream_t stream) {
  auto k = conv3x3_halo_h_kernel<CIN, COUT, EPI>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) return (int)e;
  FSD_LAUNCH(k, dim3(wgs), dim3(256), lds, stream, a);
  return (int)hipGetLastError();
}

}  // namespace

bool fsd_conv::halo_h_ok(int height, int width, int cin, int cout, int ksize) {
  static const char* env = getenv("FSD_CONV_HALO");
  if (env && env[0] == '0') return false;
  return ksize == 3 && ((cin == 32 && cout == 64) || (cin == 64 && cout == 32)) && height % kBH == 0 && width % kBW == 0;
}

int fsd_conv::halo_h_rows(int batch, int height, int width) {
  const long long blocks = (long long)batch * (height / kBH) * (width / kBW);
  return (int)(blocks < kMaxWgs ? blocks : kMaxWgs);
}

int fsd_conv::conv3x3_halo_h(const void* x, long long x_ld, const void* w_packed, int kpad, const float* bias, void* y,
                             long long y_ld, float* bn_partial, int batch, int height, int width, int cin, int cout,
                             float slope, hipStream_t stream) {
  const long long pixels = (long long)batch * height * width;
  if ((y_ld & 1) || (reinterpret_cast<uintptr_t>(y) & 3) || (x_ld & 7) || (reinterpret_cast<uintptr_t>(x) & 15) ||
      pixels >= 0x7fffffffLL || (width + 2) * x_ld >= 0x7fffffffLL || (width + 16) * y_ld >= 0x7fffffffLL)
    return FSD_ERR_UNSUPPORTED;
  HaloHArgs a;
  a.x = static_cast<const u16*>(x); a.w = static_cast<const u16*>(w_packed); a.bias = bias; a.y = static_cast<u16*>(y);
  a.bn_partial = bn_partial; a.x_ld = (unsigned)x_ld; a.y_ld = (unsigned)y_ld;
  a.H = height; a.W = width; a.Kpad = kpad;
  a.bx = width / kBW; a.by = height / kBH;
  a.blocks = batch * a.bx * a.by;
  a.slope = slope;
  const int wgs = halo_h_rows(batch, height, width);
  fsd_prof::Scope prof(fsd_prof::kGemmBf16, 2.0 * (double)pixels * cout * 9.0 * cin, stream);
  // dynamic LDS: (D + 1) patch buffers of PASSES * 4 KB, + the four waves' 4 KB store tiles for 64 outputs (see the kernel)
  const bool epi = bias != nullptr || slope != 1.f;
  if (cin == 32) {
    if ((y_ld & 7) || (reinterpret_cast<uintptr_t>(y) & 15)) return FSD_ERR_UNSUPPORTED;      // 16-byte stores
    const int lds = 5 * 3 * 4096 + 4 * 4096;
    return epi ? launch_halo_h<32, 64, true>(a, wgs, lds, stream) : launch_halo_h<32, 64, false>(a, wgs, lds, stream);
  }
  const int lds = 3 * 6 * 4096;
  return epi ? launch_halo_h<64, 32, true>(a, wgs, lds, stream) : launch_halo_h<64, 32, false>(a, wgs, lds, stream);
}
